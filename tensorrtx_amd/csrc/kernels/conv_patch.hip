// "Patch" implicit-GEMM convolution for gfx950: the input window of a 2-D output tile is staged ONCE in LDS and
// every filter tap builds its MFMA A-fragments from that LDS image, instead of re-gathering the input from
// L2/HBM once per tap (9x for a 3x3 layer) as the flat implicit-GEMM kernel (conv_igemm.hip) does.
//
//   output tile  TH x TW pixels of one image (TH*TW <= 128 = the MFMA M extent of a workgroup)
//   input patch  PH x PW = ((TH-1)*s + kh) x ((TW-1)*s + kw) pixels x CC channels (CC = 32 or 64 per chunk),
//                LDS pixel stride CC+8 halfs (16-byte pad => ds_read_b128 of 16 consecutive pixels is conflict free)
//   K loop       for chunk of CC input channels: load patch; for tap (r,q): [CC/32 MFMA k-steps]
//                weights are pre-packed in exactly that order: [Cout_pad][chunk][tap][CC_pad]
//   B tiles      (weights of one tap) are double-buffered in LDS with a register prefetch, one barrier per tap.
//   epilogue     identical to conv_igemm: +bias(BN) -> act1 -> LDS transpose -> 16-byte NHWC stores (+residual, act2)
//                into a channel slice of the destination.
//
// Same fused semantics as the reference's Conv+BN+SiLU / bottleneck chains (yolov8/src/block.cpp:79-110).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float patch_act(float v, int act, float alpha) {
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
__global__ __launch_bounds__(256) void conv_patch_f16_kernel(const ConvArgs p, const PatchGeom g) {
    constexpr int BN = 16 * NFRAG;
    constexpr int C_ROW = BN + 8;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int pstride = g.cc + 8;                 // halfs per patch pixel
    const int patch_halfs = g.ph * g.pw * pstride;
    const int b_row = g.cc + 8;                   // halfs per B row
    const int b_tile = BN * b_row;
    _Float16* patch = smem;
    _Float16* Bs = smem + patch_halfs;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tile -> (image, tile row, tile col)
    int t = blockIdx.x;
    const int tw_i = t % g.tiles_w;
    t /= g.tiles_w;
    const int th_i = t % g.tiles_h;
    const int n = t / g.tiles_h;
    const int ho0 = th_i * g.th, wo0 = tw_i * g.tw;
    const int hi0 = ho0 * p.stride_h - p.pad_h, wi0 = wo0 * p.stride_w - p.pad_w;
    const int n0 = blockIdx.y * BN;
    const _Float16* __restrict__ in = static_cast<const _Float16*>(p.in) + (size_t)n * p.H * p.W * p.ld_in;
    const _Float16* __restrict__ wgt = static_cast<const _Float16*>(p.wgt);

    // per-lane LDS offsets (halfs) of the two A fragments' rows at tap (0,0), k-step 0
    int a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int ml = wave * 32 + i * 16 + (lane & 15);
        const int mmax = g.th * g.tw - 1;
        ml = ml < mmax ? ml : mmax;  // rows beyond the tile read a valid (ignored) pixel
        const int th = ml / g.tw, tw = ml - th * g.tw;
        a_off[i] = ((th * p.stride_h) * g.pw + tw * p.stride_w) * pstride + (lane >> 4) * 8;
    }
    const int b_off = (lane & 15) * b_row + (lane >> 4) * 8;

    floatx4 acc[2][NFRAG];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int ntaps = p.kh * p.kw;
    constexpr int B_LOADS = (BN * 8 + 255) / 256;  // 16-byte pieces per thread for a 64-channel tap
    uint4 b_reg[B_LOADS];

    int kbase = 0;  // offset (halfs) of the current (chunk, tap 0) inside a packed weight row
    for (int c0 = 0; c0 < p.Cin; c0 += g.cc) {
        const int creal = (p.Cin - c0) < g.cc ? (p.Cin - c0) : g.cc;
        const int cpad = (creal + 31) / 32 * 32;  // channels of this chunk as seen by the MFMA k-steps
        const int pieces = cpad / 8;               // 16-byte pieces per pixel / per B row (4 or 8)
        const int psh = cpad == 64 ? 3 : 2;
        if (c0) __syncthreads();                   // everyone is done reading the previous chunk's patch
        // ---- stage the input patch of this channel chunk
        for (int id = tid; id < g.ph * g.pw * pieces; id += 256) {
            const int pp = id / pieces, j = id - pp * pieces;
            const int py = pp / g.pw, px = pp - py * g.pw;
            const int hi = hi0 + py, wi = wi0 + px;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && j * 8 < creal)
                v = *reinterpret_cast<const uint4*>(in + ((size_t)hi * p.W + wi) * p.ld_in + c0 + j * 8);
            *reinterpret_cast<uint4*>(patch + pp * pstride + j * 8) = v;
        }
        // ---- B tile of tap 0
        auto load_b = [&](int tap) {
#pragma unroll
            for (int q = 0; q < B_LOADS; ++q) {
                const int id = tid + q * 256;
                const int row = id >> psh, j = id & (pieces - 1);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (row < BN) v = *reinterpret_cast<const uint4*>(wgt + (size_t)(n0 + row) * p.Kpad + kbase + tap * cpad + j * 8);
                b_reg[q] = v;
            }
        };
        auto store_b = [&](int buf) {
#pragma unroll
            for (int q = 0; q < B_LOADS; ++q) {
                const int id = tid + q * 256;
                const int row = id >> psh, j = id & (pieces - 1);
                if (row < BN) *reinterpret_cast<uint4*>(Bs + buf * b_tile + row * b_row + j * 8) = b_reg[q];
            }
        };
        load_b(0);
        store_b(0);
        __syncthreads();
        int r = 0, q = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int buf = tap & 1;
            if (tap + 1 < ntaps) load_b(tap + 1);
            const _Float16* A0 = patch + (r * g.pw + q) * pstride;
            const _Float16* Bb = Bs + buf * b_tile + b_off;
            for (int ks = 0; ks < cpad; ks += 32) {
                half8 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(A0 + a_off[i] + ks);
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) {
                    const half8 bf = *reinterpret_cast<const half8*>(Bb + j * 16 * b_row + ks);
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc[i][j], 0, 0, 0);
                }
            }
            if (tap + 1 < ntaps) store_b(buf ^ 1);
            __syncthreads();
            if (++q == p.kw) {
                q = 0;
                ++r;
            }
        }
        kbase += ntaps * cpad;
    }

    // ---- epilogue (all waves passed the last barrier: LDS can be reused)
    _Float16* Cs = smem;
    {
        const int col_in = lane & 15, row_in = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const int col = j * 16 + col_in;
            const float bias = p.bias ? p.bias[n0 + col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row = wave * 32 + i * 16 + row_in + rr;
                    Cs[row * C_ROW + col] = (_Float16)patch_act(acc[i][j][rr] + bias, p.act1, p.alpha1);
                }
        }
    }
    __syncthreads();
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    const size_t img_base = (size_t)n * p.Ho * p.Wo;
    const int rows = g.th * g.tw;
    if (p.scalar_out) {
        for (int id = tid; id < rows * BN; id += 256) {
            const int row = id / BN, col = id - row * BN;
            const int th = row / g.tw, tw = row - th * g.tw;
            const int ho = ho0 + th, wo = wo0 + tw, co = n0 + col;
            if (ho < p.Ho && wo < p.Wo && co < p.Cout) {
                const size_t m = img_base + (size_t)ho * p.Wo + wo;
                float v = (float)Cs[row * C_ROW + col];
                if (res || p.act2 != ACT_NONE) v = patch_act(v + (res ? (float)res[m * p.ld_res + co] : 0.f), p.act2, p.alpha2);
                out[m * p.ld_out + co] = (_Float16)v;
            }
        }
        return;
    }
    constexpr int CHUNKS = BN / 8;
    for (int id = tid; id < rows * CHUNKS; id += 256) {
        const int row = id / CHUNKS, cc = id - row * CHUNKS;
        const int th = row / g.tw, tw = row - th * g.tw;
        const int ho = ho0 + th, wo = wo0 + tw, co = n0 + cc * 8;
        if (ho < p.Ho && wo < p.Wo && co < p.Cout) {
            const size_t m = img_base + (size_t)ho * p.Wo + wo;
            half8 v = *reinterpret_cast<const half8*>(Cs + row * C_ROW + cc * 8);
            if (res || p.act2 != ACT_NONE) {
                half8 rv = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (res) rv = *reinterpret_cast<const half8*>(res + m * p.ld_res + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (_Float16)patch_act((float)v[e] + (float)rv[e], p.act2, p.alpha2);
            }
            *reinterpret_cast<half8*>(out + m * p.ld_out + co) = v;
        }
    }
}

template <int NFRAG>
int32_t launch(const ConvArgs& a, const PatchGeom& g, hipStream_t s) {
    const int BN = 16 * NFRAG;
    const size_t lds = conv_patch_lds_bytes(a, g);
    static bool attr_set = false;  // allow > 64 KB of dynamic LDS once per instantiation
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_patch_f16_kernel<NFRAG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    dim3 grid(a.N * g.tiles_h * g.tiles_w, a.Cout_pad / BN);
    hipLaunchKernelGGL(conv_patch_f16_kernel<NFRAG>, grid, dim3(256), lds, s, a, g);
    return TRTX_OK;
}

}  // namespace

size_t conv_patch_lds_bytes(const ConvArgs& a, const PatchGeom& g) {
    const size_t main_halfs = (size_t)g.ph * g.pw * (g.cc + 8) + 2 * (size_t)a.bn * (g.cc + 8);
    const size_t epi_halfs = (size_t)128 * (a.bn + 8);
    return 2 * (main_halfs > epi_halfs ? main_halfs : epi_halfs);
}

int conv_patch_kpad(int cin, int kh, int kw, int cc) {
    int k = 0;
    for (int c0 = 0; c0 < cin; c0 += cc) {
        const int creal = cin - c0 < cc ? cin - c0 : cc;
        k += kh * kw * ((creal + 31) / 32 * 32);
    }
    return k;
}

// Pick the output tile and channel chunk for one layer; false if the layer does not fit this kernel.
bool conv_patch_plan(const ConvArgs& a, PatchGeom* g) {
    if (a.groups != 1 || a.dil_h != 1 || a.dil_w != 1 || a.Cin % 8 || a.ld_in % 8 || a.stride_h != a.stride_w ||
        a.stride_h > 2 || a.kh > 7 || a.kw > 7)
        return false;
    const int s = a.stride_h;
    double best_eff = 0;
    long best_px = 0;
    PatchGeom best{};
    for (int cc = 64; cc >= 32; cc -= 32) {
        if (cc == 64 && a.Cin <= 32) continue;
        for (int tw = 1; tw <= 128 && tw <= a.Wo; ++tw) {
            int th = 128 / tw;
            if (th > a.Ho) th = a.Ho;
            if (th < 1) continue;
            PatchGeom c{};
            c.th = th;
            c.tw = tw;
            c.cc = cc;
            c.ph = (th - 1) * s + a.kh;
            c.pw = (tw - 1) * s + a.kw;
            c.tiles_h = (a.Ho + th - 1) / th;
            c.tiles_w = (a.Wo + tw - 1) / tw;
            if (conv_patch_lds_bytes(a, c) > 64 * 1024) continue;
            const double eff = (double)a.Ho * a.Wo / ((double)c.tiles_h * c.tiles_w * 128.0);
            const long px = (long)c.ph * c.pw * c.tiles_h * c.tiles_w;  // total staged pixels (halo overhead)
            if (eff > best_eff + 1e-9 || (eff > best_eff - 1e-9 && px < best_px)) {
                best_eff = eff;
                best_px = px;
                best = c;
            }
        }
        if (best_eff > 0) break;  // prefer 64-channel chunks when any tile fits
    }
    if (best_eff < 0.5) return false;
    *g = best;
    return true;
}

int32_t conv_patch_f16(const ConvArgs& a, const PatchGeom& g, hipStream_t s) {
    switch (a.bn) {
        case 16: launch<1>(a, g, s); break;
        case 32: launch<2>(a, g, s); break;
        case 64: launch<4>(a, g, s); break;
        case 80: launch<5>(a, g, s); break;
        case 128: launch<8>(a, g, s); break;
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return check_launch("conv_patch_f16");
}

}  // namespace trtx
