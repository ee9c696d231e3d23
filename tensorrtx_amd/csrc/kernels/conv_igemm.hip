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
// per step).  LDS rows are 80 B (64 B of data + 16 B pad) to spread ds_read_b128 over the banks.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_ROW = 40;  // halfs per LDS row (32 data + 8 pad)

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case ACT_SILU: return v / (1.0f + __expf(-v));
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
    constexpr int C_ROW = BN + 8;  // halfs per row of the epilogue staging tile
    constexpr int MAIN_HALFS = 2 * (A_TILE + B_TILE);
    constexpr int EPI_HALFS = BM * C_ROW;
    constexpr int SMEM_HALFS = MAIN_HALFS > EPI_HALFS ? MAIN_HALFS : EPI_HALFS;
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
    const int kc = tid & 3;
    int a_hi0[2], a_wi0[2];
    long a_base[2];
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
        a_base[i] = (long)n * p.H * p.W;
    }
    // (r, q, c) of this thread's chunk for the current k-tile
    int kr, kq, kcin;
    {
        const int k = kc * 8;
        const int tap = k / p.Cin;
        kcin = k - tap * p.Cin;
        kr = tap / p.kw;
        kq = tap - kr * p.kw;
    }
    // B rows handled by this thread: (tid>>2) + 64*j
    constexpr int B_PASSES = (BN + 63) / 64;

    uint4 a_reg[2];
    uint4 b_reg[B_PASSES];
    const int nk = p.Kpad / BK;

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = a_hi0[i] + kr * p.dil_h;
            const int wi = a_wi0[i] + kq * p.dil_w;
            const bool ok = a_ok[i] && kr < p.kh && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (ok) v = *reinterpret_cast<const uint4*>(in + ((a_base[i] + (long)hi * p.W + wi) * p.ld_in + kcin));
            a_reg[i] = v;
        }
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) {
            const int row = (tid >> 2) + 64 * j;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < BN) v = *reinterpret_cast<const uint4*>(wgt + (size_t)(n0 + row) * p.Kpad + kt * BK + kc * 8);
            b_reg[j] = v;
        }
        // advance (r, q, c) to the next k-tile
        kcin += BK;
        while (kcin >= p.Cin) {
            kcin -= p.Cin;
            if (++kq == p.kw) {
                kq = 0;
                ++kr;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 2) + 64 * i;
            *reinterpret_cast<uint4*>(As + buf * A_TILE + row * LDS_ROW + kc * 8) = a_reg[i];
        }
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) {
            const int row = (tid >> 2) + 64 * j;
            if (row < BN) *reinterpret_cast<uint4*>(Bs + buf * B_TILE + row * LDS_ROW + kc * 8) = b_reg[j];
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
    const int frag_k = (lane >> 4) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const _Float16* Ab = As + buf * A_TILE + (wave * 32 + frag_row) * LDS_ROW + frag_k;
        const _Float16* Bb = Bs + buf * B_TILE + frag_row * LDS_ROW + frag_k;
        half8 af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(Ab + i * 16 * LDS_ROW);
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const half8 bf = *reinterpret_cast<const half8*>(Bb + j * 16 * LDS_ROW);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias + act1 -> fp16 staging tile in LDS (C/D map: col = lane&15, row = (lane>>4)*4 + r)
    _Float16* Cs = smem;
    {
        const int col_in = lane & 15;
        const int row_in = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const int col = j * 16 + col_in;
            const float bias = p.bias ? p.bias[n0 + col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wave * 32 + i * 16 + row_in + r;
                    const float v = apply_act(acc[i][j][r] + bias, p.act1, p.alpha1);
                    Cs[row * C_ROW + col] = (_Float16)v;
                }
            }
        }
    }
    __syncthreads();
    // ---- coalesced 16-byte stores (+ residual, act2) ----------------------------------------------
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    constexpr int CHUNKS_PER_ROW = BN / 8;
    constexpr int TOTAL_CHUNKS = BM * CHUNKS_PER_ROW;
    if (p.scalar_out) {
        // ragged channel counts / unaligned channel slices (e.g. 4-, 8-, 20-channel detection heads written
        // side by side into one concat buffer): element-wise stores with bounds checks
        for (int id = tid; id < BM * BN; id += 256) {
            const int row = id / BN;
            const int col = id - row * BN;
            const int m = m0 + row;
            const int co = n0 + col;
            if (m < p.M && co < p.Cout) {
                float v = (float)Cs[row * C_ROW + col];
                if (res || p.act2 != ACT_NONE) {
                    const float rv = res ? (float)res[(size_t)m * p.ld_res + co] : 0.f;
                    v = apply_act(v + rv, p.act2, p.alpha2);
                }
                out[(size_t)m * p.ld_out + co] = (_Float16)v;
            }
        }
        return;
    }
#pragma unroll
    for (int id = tid; id < TOTAL_CHUNKS; id += 256) {
        const int row = id / CHUNKS_PER_ROW;
        const int cc = id - row * CHUNKS_PER_ROW;
        const int m = m0 + row;
        const int co = n0 + cc * 8;
        if (m < p.M && co < p.Cout) {
            half8 v = *reinterpret_cast<const half8*>(Cs + row * C_ROW + cc * 8);
            if (res || p.act2 != ACT_NONE) {
                half8 rv = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (res) rv = *reinterpret_cast<const half8*>(res + (size_t)m * p.ld_res + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (_Float16)apply_act((float)v[e] + (float)rv[e], p.act2, p.alpha2);
            }
            *reinterpret_cast<half8*>(out + (size_t)m * p.ld_out + co) = v;
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
