// Fused implicit-GEMM convolution, second generation (gfx950 / MI355X):
//   * operands go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds, 16 B per lane): no VGPR staging, no ds_write;
//   * out-of-image taps, rows beyond M and K padding are handled by the buffer descriptor's bounds check (the lane
//     asks for an offset past num_records and the hardware returns zeros): the gather is branch free;
//   * 3 LDS stages, loads for k-step t+2 are issued while step t computes; waits are counted (vmcnt(L)), one raw
//     s_barrier per k-step, never a full drain inside the loop;
//   * LDS rows are 64 B, un-padded (the DMA writes lane-linear), with the 16-byte chunks XOR-swizzled on the SOURCE
//     side so that MFMA fragment reads (ds_read_b128) are bank-conflict free;
//   * MFMA operands are swapped (D^T = W * A^T) so a lane owns 4 consecutive output channels of one pixel and the
//     epilogue (bias/BN, activation, residual, activation) finishes in registers with 8-byte NHWC stores.
// Measured on the first-generation kernel (conv_igemm.hip): a k-step cost ~2000 shader cycles of which 128 were MFMA;
// the rest was exec-masked address code, the global->VGPR->LDS round trip and a full vmcnt(0) drain per step.
//
// Semantics are those of conv_igemm.hip (the reference's Conv + BN + SiLU/ReLU (+ shortcut) chains,
// yolov8/src/block.cpp:79-110, resnet/resnet50.cpp:111-151).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"
#include "kernels.h"

#ifndef TRTX_STAMP
#define TRTX_STAMP(i, kt)
#endif

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int ROW_B = 64;        // bytes per LDS row (32 halfs)
constexpr int NSTAGE = 3;
constexpr unsigned kOOB = 0x80000000u;  // offset beyond any num_records: the buffer load returns 0

__device__ __forceinline__ float act2g(float v, int act, float alpha) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        case ACT_SILU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// physical chunk = logical chunk ^ P[(row >> 2) & 3], P = {0, 2, 3, 1}: conflict-free ds_read_b128 fragments
__device__ __forceinline__ int swz4(int row) {
    return (0x1320 >> (((row >> 2) & 3) * 4)) & 3;
}

template <int NFRAG>
__global__ __launch_bounds__(256) void conv_igemm2_f16_kernel(const ConvArgs p, unsigned in_bytes, unsigned w_bytes) {
    constexpr int BN = 16 * NFRAG;
    constexpr int B_PASSES = (BN + 63) / 64;
    constexpr int A_BYTES = BM * ROW_B;                       // 8 KB
    constexpr int B_BYTES = B_PASSES * 64 * ROW_B;            // 4 or 8 KB (rows beyond BN are dummy targets)
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int LOADS_PER_TILE = 2 + B_PASSES;              // buffer_load...lds instructions per wave per k-step
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, w_bytes, 0x00020000);

    // ---- per-lane source description.  Lane l of a wave fills LDS row 16*wave + (l >> 2) (+64 in pass 1), physical
    // chunk (l & 3); it therefore fetches LOGICAL chunk (l & 3) ^ swz(row).
    const int lrow = lane >> 2;                       // 0..15
    const int lchunk = (lane & 3) ^ swz4(lrow);       // (16*wave is a multiple of 16: swz depends on lrow only)
    int a_hi0[2], a_wi0[2];
    unsigned a_base[2];  // byte offset of (n, hi0, wi0, channel lchunk*8); wraps for border pixels, masked below
    bool a_ok[2];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 64 * i + 16 * wave + lrow;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        a_hi0[i] = ho * p.stride_h - p.pad_h;
        a_wi0[i] = wo * p.stride_w - p.pad_w;
        a_base[i] = (unsigned)((((long)n * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.ld_in + lchunk * 8) * 2u;
    }
    unsigned b_base[B_PASSES];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
        const int row = 64 * j + 16 * wave + lrow;
        b_base[j] = row < BN ? (unsigned)(((size_t)(n0 + row) * p.Kpad + lchunk * 8) * 2) : kOOB;
    }
    const bool uniform_taps = (p.Cin % BK) == 0;
    int ur = 0, uq = 0, uc = 0;  // wave-uniform tap / channel of the k-step being LOADED
    int kr, kq, kcin;            // per-lane position (general path)
    {
        const int k = lchunk * 8;
        const int tap = k / p.Cin;
        kcin = k - tap * p.Cin;
        kr = tap / p.kw;
        kq = tap - kr * p.kw;
    }

    auto issue_tile = [&](int kt, int stage) {
        char* sbase = smem + stage * STAGE_BYTES;
        unsigned voff[2];
        if (uniform_taps) {
            const unsigned toff = (unsigned)(((ur * p.dil_h * p.W + uq * p.dil_w) * p.ld_in + uc) * 2);
            const bool tap_ok = ur < p.kh;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hi = a_hi0[i] + ur * p.dil_h, wi = a_wi0[i] + uq * p.dil_w;
                const bool ok = a_ok[i] && tap_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                voff[i] = ok ? a_base[i] + toff : kOOB;
            }
            uc += BK;
            if (uc >= p.Cin) {
                uc = 0;
                if (++uq == p.kw) {
                    uq = 0;
                    ++ur;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hi = a_hi0[i] + kr * p.dil_h, wi = a_wi0[i] + kq * p.dil_w;
                const bool ok = a_ok[i] && kr < p.kh && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const unsigned toff = (unsigned)(((kr * p.dil_h * p.W + kq * p.dil_w) * p.ld_in + kcin - lchunk * 8) * 2);
                voff[i] = ok ? a_base[i] + toff : kOOB;
            }
            kcin += BK;
            while (kcin >= p.Cin) {
                kcin -= p.Cin;
                if (++kq == p.kw) {
                    kq = 0;
                    ++kr;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sbase + (64 * i + 16 * wave) * ROW_B), 16, voff[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) {
            const unsigned bo = b_base[j] == kOOB ? kOOB : b_base[j] + (unsigned)(kt * BK * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sbase + A_BYTES + (64 * j + 16 * wave) * ROW_B), 16, bo, 0, 0, 0);
        }
    };

    floatx4 acc[2][NFRAG];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.Kpad / BK;
    issue_tile(0, 0);
    if (nk > 1) issue_tile(1, 1);

    // fragment read offsets (bytes inside a stage): row (lane & 15), logical chunk (lane >> 4)
    const int frow = lane & 15;
    const int fchunk = ((lane >> 4) ^ swz4(frow)) * 16;
    const int a_frag = (wave * 32 + frow) * ROW_B + fchunk;
    const int b_frag = A_BYTES + frow * ROW_B + fchunk;

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        TRTX_STAMP(0, kt);
        // tile kt has landed once at most the LOADS_PER_TILE loads of tile kt+1 are still in flight
        if (kt + 1 < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_TILE) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRTX_STAMP(1, kt);
        __builtin_amdgcn_s_barrier();
        TRTX_STAMP(2, kt);
        if (kt + 2 < nk) {
            int s2 = stage + 2;
            s2 = s2 >= NSTAGE ? s2 - NSTAGE : s2;
            issue_tile(kt + 2, s2);
        }
        TRTX_STAMP(3, kt);
        const char* sb = smem + stage * STAGE_BYTES;
        half8 af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(sb + a_frag + i * 16 * ROW_B);
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const half8 bf = *reinterpret_cast<const half8*>(sb + b_frag + j * 16 * ROW_B);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, af[i], acc[i][j], 0, 0, 0);
        }
        TRTX_STAMP(4, kt);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }

    // ---- epilogue in registers: lane owns channels n0 + 16j + 4*(lane>>4) + [0,4) of pixel m0 + 32*wave + 16i + (lane&15)
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    const int px_in = lane & 15;
    const int ch_in = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wave * 32 + i * 16 + px_in;
        if (m >= p.M) continue;
        _Float16* orow = out + (size_t)m * p.ld_out;
        const _Float16* rrow = res ? res + (size_t)m * p.ld_res : nullptr;
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const int co = n0 + j * 16 + ch_in;
            if (co >= p.Cout) continue;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + co);
            float v[4];
            v[0] = act2g(acc[i][j][0] + bv.x, p.act1, p.alpha1);
            v[1] = act2g(acc[i][j][1] + bv.y, p.act1, p.alpha1);
            v[2] = act2g(acc[i][j][2] + bv.z, p.act1, p.alpha1);
            v[3] = act2g(acc[i][j][3] + bv.w, p.act1, p.alpha1);
            if (!p.scalar_out) {
                if (res || p.act2 != ACT_NONE) {
                    half4 rv = half4{0, 0, 0, 0};
                    if (res) rv = *reinterpret_cast<const half4*>(rrow + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act2g((float)(_Float16)v[e] + (float)rv[e], p.act2, p.alpha2);
                }
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                *reinterpret_cast<half4*>(orow + co) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (co + e < p.Cout) {
                        float x = v[e];
                        if (res || p.act2 != ACT_NONE) x = act2g((float)(_Float16)x + (res ? (float)rrow[co + e] : 0.f), p.act2, p.alpha2);
                        orow[co + e] = (_Float16)x;
                    }
                }
            }
        }
    }
}

template <int NFRAG>
void launch2(const ConvArgs& a, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    const int BN = 16 * NFRAG;
    dim3 grid((a.M + BM - 1) / BM, a.Cout_pad / BN);
    hipLaunchKernelGGL(conv_igemm2_f16_kernel<NFRAG>, grid, dim3(256), 0, s, a, in_bytes, w_bytes);
}

}  // namespace

// The buffer descriptors need 32-bit byte extents: input slice and packed weights must each stay below 2 GB.
bool conv_igemm2_supported(const ConvArgs& a) {
    if (!conv_igemm_supported(a)) return false;
    const double in_b = ((double)a.N * a.H * a.W * a.ld_in) * 2.0, w_b = (double)a.Cout_pad * a.Kpad * 2.0;
    return in_b < 2.0e9 && w_b < 2.0e9 && a.stride_h * 1 > 0;
}

int32_t conv_igemm2_f16(const ConvArgs& a, hipStream_t s) {
    if (!conv_igemm2_supported(a)) return TRTX_ERR_UNSUPPORTED;
    // extent of the addressed slice: last pixel's first byte + the channels this conv reads
    const unsigned in_bytes = (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ld_in + a.Cin) * 2);
    const unsigned w_bytes = (unsigned)((size_t)a.Cout_pad * a.Kpad * 2);
    switch (a.bn) {
        case 16: launch2<1>(a, in_bytes, w_bytes, s); break;
        case 32: launch2<2>(a, in_bytes, w_bytes, s); break;
        case 64: launch2<4>(a, in_bytes, w_bytes, s); break;
        case 80: launch2<5>(a, in_bytes, w_bytes, s); break;
        case 128: launch2<8>(a, in_bytes, w_bytes, s); break;
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return check_launch("conv_igemm2_f16");
}

}  // namespace trtx
