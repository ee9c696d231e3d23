// Fused implicit-GEMM convolution for gfx950 (MI355X): NHWC fp16 in/out, fp32 accumulate on the
// matrix cores (v_mfma_f32_16x16x32_f16), epilogue = +bias(folded BN) -> act1 -> +residual -> act2,
// stored through an LDS transpose as 16-byte NHWC rows into an arbitrary channel slice of the
// destination (so addConcatenation costs nothing).
//
// This is the engine side of what the reference delegates to TensorRT for every
// addConvolutionNd + addScale(BN) + addActivation/addElementWise chain
// (yolov8/src/block.cpp:79-96 convBnSiLU; resnet/resnet50.cpp:111-151 bottleneck).
//
// GEMM view:  M = N*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin (k = (r*kw+q)*Cin + c).
//   A[m][k] gathered on the fly from the NHWC input (zero outside the image),
//   B[n][k] = pre-packed weights [Cout_pad][K_pad] (K contiguous, zero padded).
// Tile: 128 pixels x (16*NFRAG) channels x 32 k per step; 4 waves, wave w owns pixel rows
// [32w, 32w+32) x all columns (2 x NFRAG accumulator fragments).  Global loads for step t+1 are
// issued before the MFMAs of step t and written to the other LDS buffer afterwards (one barrier
// per step).  LDS rows are 64 B with an XOR chunk swizzle so fragment reads are bank-conflict free.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"
#include "kernels.h"

// development hook: tools/hip/igemm_phase_probe.hip defines TRTX_STAMP to record shader-clock stamps of one k-step
#ifndef TRTX_STAMP
#define TRTX_STAMP(i, kt)
#endif

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_ROW = 32;  // halfs per LDS row: 64 B, un-padded; the four 16-byte chunks of a row are XOR-swizzled

// chunk permutation that makes ds_read_b128 of an MFMA fragment (16 rows x one chunk per 16-lane service group, see the
// group table in MI355X_MICROARCH.md) hit 16 distinct 16-byte slots: physical chunk = logical chunk ^ P[(row >> 2) & 3]
__device__ __forceinline__ int swz(int row) {
    return (0x1320 >> (((row >> 2) & 3) * 4)) & 3;  // P = {0, 2, 3, 1}
}

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        case ACT_SILU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));  // v_exp + v_rcp (1 ulp), rounded to fp16 afterwards
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

template <int NFRAG>
__global__ __launch_bounds__(256) void conv_igemm_f16_kernel(const ConvArgs p) {
    constexpr int BN = 16 * NFRAG;
    constexpr int A_TILE = BM * LDS_ROW;  // halfs
    constexpr int B_TILE = BN * LDS_ROW;
    constexpr int SMEM_HALFS = 2 * (A_TILE + B_TILE);
    __shared__ __attribute__((aligned(16))) _Float16 smem[SMEM_HALFS];
    _Float16* As = smem;
    _Float16* Bs = smem + 2 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const _Float16* __restrict__ in = static_cast<const _Float16*>(p.in);
    const _Float16* __restrict__ wgt = static_cast<const _Float16*>(p.wgt);

    // ---- per-thread A-gather state: rows (tid>>2) and (tid>>2)+64, k-chunk (tid&3) --------------
    // Address generation is kept off the vector ALU as far as possible (it, not the MFMA pipe, was the busiest
    // unit of the first version): each thread precomputes one pointer per row; the filter tap (r, q) and the
    // channel offset of a k-tile are wave-uniform, so their offset is a scalar added per step.
    const int kc = tid & 3;
    int a_hi0[2], a_wi0[2];
    const _Float16* a_ptr[2];
    bool a_ok[2];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        a_hi0[i] = ho * p.stride_h - p.pad_h;
        a_wi0[i] = wo * p.stride_w - p.pad_w;
        // pointer to (n, hi0, wi0, channel kc*8); may lie outside the tensor for border pixels, it is only
        // dereferenced after the bounds test
        a_ptr[i] = in + (((long)n * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.ld_in + kc * 8;
    }
    constexpr int B_PASSES = (BN + 63) / 64;
    const _Float16* b_ptr = wgt + (size_t)(n0 + (tid >> 2)) * p.Kpad + kc * 8;  // pass j adds 64*j rows
    const bool uniform_taps = (p.Cin % BK) == 0;  // every k-tile lies inside one filter tap
    // wave-uniform position of the current k-tile: tap (ur, uq), first channel uc
    int ur = 0, uq = 0, uc = 0;
    // per-thread position (general path: Cin % 32 != 0, a k-tile may straddle taps)
    int kr, kq, kcin;
    {
        const int k = kc * 8;
        const int tap = k / p.Cin;
        kcin = k - tap * p.Cin;
        kr = tap / p.kw;
        kq = tap - kr * p.kw;
    }

    uint4 a_reg[2];
    uint4 b_reg[B_PASSES];
    const int nk = p.Kpad / BK;

    auto load_tile = [&](int kt) {
        if (uniform_taps) {
            const int toff = (ur * p.dil_h * p.W + uq * p.dil_w) * p.ld_in + uc;  // scalar
            const bool tap_ok = ur < p.kh;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hi = a_hi0[i] + ur * p.dil_h;
                const int wi = a_wi0[i] + uq * p.dil_w;
                const bool ok = a_ok[i] && tap_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (ok) v = *reinterpret_cast<const uint4*>(a_ptr[i] + toff);
                a_reg[i] = v;
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
                const int hi = a_hi0[i] + kr * p.dil_h;
                const int wi = a_wi0[i] + kq * p.dil_w;
                const bool ok = a_ok[i] && kr < p.kh && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (ok) v = *reinterpret_cast<const uint4*>(a_ptr[i] + ((long)(kr * p.dil_h) * p.W + kq * p.dil_w) * p.ld_in + kcin - kc * 8);
                a_reg[i] = v;
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
        for (int j = 0; j < B_PASSES; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((tid >> 2) + 64 * j < BN) v = *reinterpret_cast<const uint4*>(b_ptr + (size_t)(64 * j) * p.Kpad + kt * BK);
            b_reg[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 2) + 64 * i;
            *reinterpret_cast<uint4*>(As + buf * A_TILE + row * LDS_ROW + (kc ^ swz(row)) * 8) = a_reg[i];
        }
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) {
            const int row = (tid >> 2) + 64 * j;
            if (row < BN) *reinterpret_cast<uint4*>(Bs + buf * B_TILE + row * LDS_ROW + (kc ^ swz(row)) * 8) = b_reg[j];
        }
    };

    floatx4 acc[2][NFRAG];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frag_row = lane & 15;
    const int frag_k = ((lane >> 4) ^ swz(frag_row)) * 8;  // tile bases are multiples of 16 rows: swz depends on the lane only
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        TRTX_STAMP(0, kt);
        if (kt + 1 < nk) load_tile(kt + 1);
        TRTX_STAMP(1, kt);
        const _Float16* Ab = As + buf * A_TILE + (wave * 32 + frag_row) * LDS_ROW + frag_k;
        const _Float16* Bb = Bs + buf * B_TILE + frag_row * LDS_ROW + frag_k;
        half8 af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(Ab + i * 16 * LDS_ROW);
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const half8 bf = *reinterpret_cast<const half8*>(Bb + j * 16 * LDS_ROW);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, af[i], acc[i][j], 0, 0, 0);  // D^T: rows = channels
        }
        TRTX_STAMP(2, kt);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        TRTX_STAMP(3, kt);
        __syncthreads();
        TRTX_STAMP(4, kt);
    }

    // ---- epilogue --------------------------------------------------------------------------------------
    // The MFMAs were issued with the operands swapped (D^T = W * A^T), so each lane holds, per fragment, FOUR
    // CONSECUTIVE OUTPUT CHANNELS of ONE pixel: channel = n0 + 16j + 4*(lane>>4) + r, pixel = m0 + 32*wave + 16i +
    // (lane&15).  They are finished in registers (bias, activation, residual, activation) and stored straight to
    // NHWC global memory as 8-byte runs: no LDS staging, no barrier, 4x fewer store instructions than the first
    // version (instruction issue, not bandwidth, bounds these layers).
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
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
            float v[4];
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + co);  // bias is padded to Cout_pad
            v[0] = apply_act(acc[i][j][0] + bv.x, p.act1, p.alpha1);
            v[1] = apply_act(acc[i][j][1] + bv.y, p.act1, p.alpha1);
            v[2] = apply_act(acc[i][j][2] + bv.z, p.act1, p.alpha1);
            v[3] = apply_act(acc[i][j][3] + bv.w, p.act1, p.alpha1);
            if (!p.scalar_out) {
                if (res || p.act2 != ACT_NONE) {
                    half4 rv = half4{0, 0, 0, 0};
                    if (res) rv = *reinterpret_cast<const half4*>(rrow + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act((float)(_Float16)v[e] + (float)rv[e], p.act2, p.alpha2);
                }
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                *reinterpret_cast<half4*>(orow + co) = o;
            } else {
                // ragged channel counts / unaligned channel slices: element-wise with bounds checks
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (co + e < p.Cout) {
                        float x = v[e];
                        if (res || p.act2 != ACT_NONE) x = apply_act((float)(_Float16)x + (res ? (float)rrow[co + e] : 0.f), p.act2, p.alpha2);
                        orow[co + e] = (_Float16)x;
                    }
                }
            }
        }
    }
}

template <int NFRAG>
void launch(const ConvArgs& a, hipStream_t s) {
    const int BN = 16 * NFRAG;
    dim3 grid((a.M + BM - 1) / BM, a.Cout_pad / BN);
    hipLaunchKernelGGL(conv_igemm_f16_kernel<NFRAG>, grid, dim3(256), 0, s, a);
}

}  // namespace

int conv_igemm_pick_bn(int cout) {
    // widest tile that divides the padded Cout without waste
    if (cout % 128 == 0) return 128;
    if (cout % 80 == 0) return 80;
    if (cout % 64 == 0) return 64;
    if (cout % 32 == 0) return 32;
    if (cout % 16 == 0) return 16;
    const int pad16 = (cout + 15) / 16 * 16;
    return conv_igemm_pick_bn(pad16);
}

bool conv_igemm_supported(const ConvArgs& a) {
    const bool out_vec = a.ld_out % 8 == 0 && a.Cout % 8 == 0 && (!a.residual || a.ld_res % 8 == 0);
    return a.Cin % 8 == 0 && a.ld_in % 8 == 0 && a.groups == 1 && a.Kpad % BK == 0 && (out_vec || a.scalar_out);
}

int32_t conv_igemm_f16(const ConvArgs& a, hipStream_t s) {
    if (!conv_igemm_supported(a)) return TRTX_ERR_UNSUPPORTED;
    switch (a.bn) {
        case 16: launch<1>(a, s); break;
        case 32: launch<2>(a, s); break;
        case 64: launch<4>(a, s); break;
        case 80: launch<5>(a, s); break;
        case 128: launch<8>(a, s); break;
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return check_launch("conv_igemm_f16");
}

}  // namespace trtx
