// Fused implicit-GEMM convolution for gfx950 (MI355X): NHWC fp16 activations, fp16 weights packed [Cout_pad][Kpad],
// fp32 accumulation on v_mfma_f32_16x16x32_f16, epilogue bias(BN) -> act1 -> (+residual) -> act2 in registers.
// This is the kernel behind every Conv(+Scale)(+Activation)(+ElementWise SUM)(+Activation) chain the reference
// builders emit (yolov8/src/block.cpp:79-110, resnet/resnet50.cpp:111-151, rcnn/backbone.hpp:104-169).
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = taps x CinK with CinK = Cin rounded up to the k-step (32 or 64
// halfs), so a k-step never straddles a filter tap and every address decision is wave-uniform.
//
// What the measurements on MI355X dictated (tools/hip/ldsdma_bw.hip, tools/pmc_conv2.sh, DESIGN.md "conv kernel"):
//   * the loop was instruction-issue bound (~150 instructions and ~20 branches per k-step for 8 MFMAs), not memory
//     bound: L2 hit rate 93 %, fabric reads 1x the input, 60 % of wave time in s_waitcnt.  The steady-state k-step
//     here is straight-line code: tap validity comes from a per-pixel bit mask built once, the tap walk is scalar
//     select arithmetic, the three pipeline stages are unrolled so every LDS address is an immediate, and tiles past
//     the end of K are issued as out-of-range (zero-fill, no memory access) instead of being branched around;
//   * operands go L2 -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction): out-of-image taps,
//     rows >= M and channels >= Cin are handled by the buffer descriptor's range check (offset 0x80000000 -> zeros);
//   * the vector L1 serves whole 128-B lines: a 64-B slice of a pixel costs the same slot as the full line, so layers
//     with Cin % 64 == 0 use 64-wide k-steps (BKT = 64: one line per pixel per step, half the barriers);
//   * LDS rows are un-padded (the DMA writes lane-linear) and the 16-byte chunks are XOR-swizzled on the SOURCE side so
//     the MFMA fragment reads (ds_read_b128) are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0);
//   * consecutive workgroup ids are dealt round-robin to the 8 XCDs, so ids are remapped to give every XCD a contiguous
//     range of output tiles (halo rows and the A tile shared by n-tiles stay in one L2): fabric reads 2.3x -> 1.0x.
//
// This header holds the DEVICE code (tile function, epilogues, the single-problem and the grouped kernel entry points) and is included by the three
// translation units that instantiate it: conv_igemm.hip (fp16 and - I8 - int8 operands), conv_igemm_f32.hip (F32: fp32 operands on v_mfma_f32_16x16x4_f32,
// round 5) and, through patch_tile.h, the resident-patch 3x3 kernels.  Template switches of conv_igemm_tile, all compile-time: NFRAG / MI tile width and
// height, BKT / TPS k-step width and taps per step, I8 / F32 operand type, RS operands through registers instead of LDS-DMA, UP folded nearest upsample,
// ONE plain-GEMM addressing, ROLES fetching + multiplying waves (instantiated for F32), WN / NW / NSTO / PRE wave grid, waves, stages, read-ahead.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "../common.h"
#include "kernels.h"
#include "launch.h"
#include "patch_index.h"

#ifndef TRTX_STAMP
#define TRTX_STAMP(i, kt)
#endif
#ifndef TRTX_MARK   // launch anatomy (tools/hip/igemm_launch_anatomy.hip): 0 entry, 1 first tile issued, 2 k-loop done, 3 epilogue done
#define TRTX_MARK(i)
#endif
// Ablation switches for the timing experiments of tools/conv_dbg.sh (build with -DTRTX_CONV_ABLATE to get them from the
// TRTX_CONV_DBG environment variable: 1 A loads range-checked away, 2 B loads, 4 no ds_read/MFMA, 8 no epilogue, 16 no
// k-loop).  In the product build the flag word is the constant 0 and every test on it folds away.
#ifdef TRTX_CONV_ABLATE
#define TRTX_DBG(flags) (flags)
#else
#define TRTX_DBG(flags) 0
#endif

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int NSTAGE = 3;
constexpr unsigned kOOB = 0x80000000u;  // offset beyond any num_records: the buffer load returns 0
constexpr int kMaxTaps = 30;            // tap-validity mask is one 32-bit word (+2 bits of run-out past the last tap)

__device__ __attribute__((noinline)) float act_slow(float v, int act, float alpha) {
    switch (act) {
        case ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        case ACT_MISH: return mish_ref(v);
        default: return v;
    }
}
__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    if (act == ACT_NONE) return v;
    if (act == ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    return act_slow(v, act, alpha);
}

// bit t set iff 0 <= x0 + t < extent, for t in [0, k), k <= 30
__device__ __forceinline__ unsigned tap_range_mask(int x0, int k, int extent) {
    const int lo = x0 < 0 ? -x0 : 0;
    int hi = extent - x0;
    hi = hi < k ? hi : k;
    return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
}

// physical 16-byte chunk = logical chunk ^ swz(row); see the header comment
template <int BKT>
__device__ __forceinline__ int swz(int row) {
    if (BKT == 32) return (0x1320 >> (((row >> 2) & 3) * 4)) & 3;  // P = {0, 2, 3, 1} over (row >> 2) & 3
    return (row >> 1) & 7;
}

// The item loop of conv_epilogue's common case (see there), with (RES) or without a residual.
template <int NFRAG, int MI, bool I8, bool RES, int PS, int BIAS_LDS_ON, typename PixelOf>
__device__ __forceinline__ void conv_epilogue_fast(const ConvArgs& p, floatx4 (&acc)[MI][NFRAG], intx4 (&acci)[MI][NFRAG], char* mine, const char* bias_lds, int lane,
                                                   int n0, int rbase, float4 bf0, float4 bf1, bool second, PixelOf&& pixel_of) {
    constexpr int BN = 16 * NFRAG, CPR = BN / 8, ITEMS = 16 * CPR, NIT = (ITEMS + 63) / 64;
    constexpr bool BIAS_FIXED = !BIAS_LDS_ON;
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    const int px_in = lane & 15;
    const int ch_in = (lane >> 4) * 4;
    const half8 zero8 = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma nounroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int ii = 0; ii < MI; ++ii) {
            if (ii != i) continue;
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) {
                floatx4 v = acc[ii][j];
                if constexpr (I8) {
                    const float4 cs = *reinterpret_cast<const float4*>(p.cscale + n0 + j * 16 + ch_in);
                    v = floatx4{(float)acci[ii][j][0] * cs.x, (float)acci[ii][j][1] * cs.y, (float)acci[ii][j][2] * cs.z, (float)acci[ii][j][3] * cs.w};
                }
                *reinterpret_cast<floatx4*>(mine + px_in * PS + (j * 16 + ch_in) * 4) = v;
            }
        }
        // item (q -> row, 8-channel column, pixel).  With a residual: its 16 bytes are fetched one item ahead, after the previous item's
        // store, by an unconditional load from a clamped address (a conditional one would wait where it stands)
        auto locate = [&](int q, int& row, int& cc, int& m) {
            row = q / CPR, cc = q % CPR;
            m = q < ITEMS ? pixel_of(rbase + i * 16 + row) : -1;
            return m >= 0 && n0 + cc * 8 < p.Cout;
        };
        int row, cc, m;
        bool ok = locate(lane, row, cc, m);
        half8 rv = zero8;
        if constexpr (RES) rv = *reinterpret_cast<const half8*>(res + (ok ? (size_t)m * p.ld_res + n0 + cc * 8 : (size_t)0));
#pragma nounroll
        for (int it = 0; it < NIT; ++it) {
            half8 v = zero8;
            if (ok) {
                const floatx4 lo = *reinterpret_cast<const floatx4*>(mine + row * PS + cc * 32);
                const floatx4 hi = *reinterpret_cast<const floatx4*>(mine + row * PS + cc * 32 + 16);
                float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (p.bias) {
                    float4 b0 = bf0, b1 = bf1;
                    if constexpr (!BIAS_FIXED) {
                        b0 = *reinterpret_cast<const float4*>(bias_lds + cc * 32);
                        b1 = *reinterpret_cast<const float4*>(bias_lds + cc * 32 + 16);
                    }
                    x[0] += b0.x; x[1] += b0.y; x[2] += b0.z; x[3] += b0.w;
                    x[4] += b1.x; x[5] += b1.y; x[6] += b1.z; x[7] += b1.w;
                }
                if (p.act1 == ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-x[e])));
                } else if (p.act1 == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e] > 0.f ? x[e] : 0.f);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e]);
                }
                if (second) {
                    if (p.act2 == ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = round_to_half((float)v[e] + (float)rv[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t = (float)v[e] + (float)rv[e];
                            v[e] = round_to_half(t > 0.f ? t : 0.f);
                        }
                    }
                }
                *reinterpret_cast<half8*>(out + (size_t)m * p.ld_out + n0 + cc * 8) = v;
            }
            if (it + 1 < NIT) {
                ok = locate(lane + (it + 1) * 64, row, cc, m);
                if constexpr (RES) rv = *reinterpret_cast<const half8*>(res + (ok ? (size_t)m * p.ld_res + n0 + cc * 8 : (size_t)0));
            }
        }
    }
}

// ---- epilogue shared by the implicit-GEMM kernels.  The accumulators (lane owns channels 16j + 4*(lane>>4) + [0,4) of pixel
// 16i + (lane&15) of its wave's WR rows) go through a wave-private fp32 LDS tile, 16 rows at a time, and come back row-major:
// one lane = 8 consecutive channels of one pixel, so residual reads and output stores are whole 16-byte chunks (128-byte lines
// per 8 lanes) and - this is the point - the code that finishes them (bias, act1, rounding, residual, act2, requantisation, ragged
// stores) exists ONCE, in a rolled loop with wave-uniform branches, instead of once per accumulator fragment and activation
// kind.  Unrolled, that code was 80 % of a 45-90 KB kernel against a 64 KB instruction cache shared by two CUs
// (profiles/r02_code_size.txt).  `pixel_of(t)` maps row t of the tile (0 .. 64 * MI) to the output pixel index, or -1.
template <int NFRAG, int MI, bool I8, int LDS_BYTES, int NWAVES = 4, typename PixelOf>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, floatx4 (&acc)[MI][NFRAG], intx4 (&acci)[MI][NFRAG], char* smem, int wave, int lane,
                                              int n0, PixelOf&& pixel_of, int row0 = -1) {
    constexpr int BN = 16 * NFRAG;
    constexpr int WR = 16 * MI;
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    const int px_in = lane & 15;
    const int ch_in = (lane >> 4) * 4;
    const bool second = res || p.act2 != ACT_NONE;
    constexpr int PS = BN * 4 + 16;   // fp32 row stride of the staging tile (padded: 16 consecutive rows start in distinct bank groups)
    static_assert(NWAVES * 16 * PS <= LDS_BYTES, "epilogue tile must fit in the stage buffers");
    constexpr int CPR = BN / 8;       // 8-channel items per row
    constexpr int ITEMS = 16 * CPR;   // items of one 16-row slab of this wave
    // A lane's items q = lane, lane + 64, ... all sit in the same 8-channel column when CPR divides 64 (every tile width but 80): its bias
    // is fetched ONCE, here, and the L2 round trip passes under the barrier and the staging writes below instead of standing in front of the
    // first store of every slab (round 4).
    // The 80-wide tile (CPR = 10) has no such column: its wave fetches the tile's 80 bias values once, four per lane, and keeps them in a
    // private LDS strip behind the staging tiles; an item reads its eight from there (an LDS read where an L2 round trip stood).
    constexpr bool BIAS_FIXED = (64 % CPR) == 0;
    constexpr int BIAS_LDS = BIAS_FIXED ? 0 : BN * 4;
    static_assert(NWAVES * (16 * PS + BIAS_LDS) <= LDS_BYTES, "epilogue tile + bias strip must fit in the stage buffers");
    float4 bf0 = make_float4(0.f, 0.f, 0.f, 0.f), bf1 = bf0;
    if (BIAS_FIXED && p.bias) {
        const int co0 = n0 + (lane % CPR) * 8;
        const int cc0 = co0 < p.Cout ? co0 : 0;   // (clamped, not conditional: a conditional load waits where it stands)
        bf0 = *reinterpret_cast<const float4*>(p.bias + cc0);
        bf1 = *reinterpret_cast<const float4*>(p.bias + cc0 + 4);
    }
    if (!BIAS_FIXED && p.bias) bf0 = *reinterpret_cast<const float4*>(p.bias + n0 + (lane < BN / 4 ? lane * 4 : 0));   // (n0 + BN <= Cout_pad)
    __syncthreads();  // every wave is done reading the last stage
    char* mine = smem + wave * 16 * PS;
    char* bias_lds = smem + NWAVES * 16 * PS + wave * BIAS_LDS;
    if (!BIAS_FIXED && p.bias && lane < BN / 4) *reinterpret_cast<float4*>(bias_lds + lane * 16) = bf0;   // wave-private: ordered with this wave's reads
    const int rbase = row0 < 0 ? wave * WR : row0;   // first tile row of this wave (waves may also be split along N: WN below)
    // The common case - 16-byte fp16 stores, ReLU / SiLU / no activation, fp16 residual - has its own item loop (round 4).  In the general
    // loop below the rare paths (int8 residual, element-wise ragged loads and stores, the slow activations) put global LOADS into the loop
    // body; the compiler's wait for them (`s_waitcnt vmcnt(0)`, merged at the loop header) then also waits for the previous item's STORE
    // to be acknowledged: 4-8 store round trips in series per wave, 2.3-3.9 us per tile measured (profiles/r04_launch_anatomy.txt).  The
    // fast loop holds no load but the residual's, and fetches that one item ahead, so the wait it needs leaves the stores in flight.
    const bool fast = !p.scalar_out && !p.out_i8 && !(res && p.res_i8) && (p.act1 == ACT_SILU || p.act1 == ACT_RELU || p.act1 == ACT_NONE) &&
                      (p.act2 == ACT_NONE || p.act2 == ACT_RELU);
#ifndef TRTX_NO_FAST_EPILOGUE   // (A/B builds of the round-4 measurement only)
    if (fast) {
        // two copies of the loop, with and without a residual: the one without holds no global load at all, so no wait for one either
        if (res) conv_epilogue_fast<NFRAG, MI, I8, true, PS, !BIAS_FIXED>(p, acc, acci, mine, bias_lds, lane, n0, rbase, bf0, bf1, second, pixel_of);
        else conv_epilogue_fast<NFRAG, MI, I8, false, PS, !BIAS_FIXED>(p, acc, acci, mine, bias_lds, lane, n0, rbase, bf0, bf1, second, pixel_of);
        return;
    }
#endif
#pragma nounroll
    for (int i = 0; i < MI; ++i) {
        // accumulators of row slab i -> LDS (int8: dequantised by input scale * weight scale of the channel)
#pragma unroll
        for (int ii = 0; ii < MI; ++ii) {
            if (ii != i) continue;
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) {
                floatx4 v = acc[ii][j];
                if constexpr (I8) {
                    const float4 cs = *reinterpret_cast<const float4*>(p.cscale + n0 + j * 16 + ch_in);
                    v = floatx4{(float)acci[ii][j][0] * cs.x, (float)acci[ii][j][1] * cs.y, (float)acci[ii][j][2] * cs.z, (float)acci[ii][j][3] * cs.w};
                }
                *reinterpret_cast<floatx4*>(mine + px_in * PS + (j * 16 + ch_in) * 4) = v;
            }
        }
        // wave-private tile: the LDS accesses of one wave are ordered, no barrier needed
#pragma nounroll
        for (int q = lane; q < ITEMS; q += 64) {
            const int row = q / CPR, cc = q % CPR;
            const int m = pixel_of(rbase + i * 16 + row);
            const int co = n0 + cc * 8;
            if (m < 0 || co >= p.Cout) continue;
            const floatx4 lo = *reinterpret_cast<const floatx4*>(mine + row * PS + cc * 32);
            const floatx4 hi = *reinterpret_cast<const floatx4*>(mine + row * PS + cc * 32 + 16);
            float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (p.bias) {
                float4 b0 = bf0, b1 = bf1;
                if constexpr (!BIAS_FIXED) {
                    b0 = *reinterpret_cast<const float4*>(bias_lds + cc * 32);
                    b1 = *reinterpret_cast<const float4*>(bias_lds + cc * 32 + 16);
                }
                x[0] += b0.x; x[1] += b0.y; x[2] += b0.z; x[3] += b0.w;
                x[4] += b1.x; x[5] += b1.y; x[6] += b1.z; x[7] += b1.w;
            }
            half8 v;
            if (p.act1 == ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-x[e])));
            } else if (p.act1 == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e] > 0.f ? x[e] : 0.f);
            } else if (p.act1 == ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e]);
            } else {
#pragma nounroll
                for (int e = 0; e < 8; ++e) v[e] = round_to_half(act_slow(x[e], p.act1, p.alpha1));
            }
            const bool vec = !p.scalar_out;
            if (second) {
                half8 rv = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (res) {
                    if (p.res_i8) {  // int8 residual: 8 bytes, dequantised with its tensor scale
                        const long long rq = *reinterpret_cast<const long long*>(static_cast<const int8_t*>(p.residual) + (size_t)m * p.ld_res + co);
#pragma unroll
                        for (int e = 0; e < 8; ++e) rv[e] = round_to_half((float)(int8_t)(rq >> (8 * e)) * p.res_scale);
                    } else if (vec) {
                        rv = *reinterpret_cast<const half8*>(res + (size_t)m * p.ld_res + co);
                    } else {
#pragma nounroll
                        for (int e = 0; e < 8; ++e)
                            if (co + e < p.Cout) rv[e] = res[(size_t)m * p.ld_res + co + e];
                    }
                }
                if (p.act2 == ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_to_half((float)v[e] + (float)rv[e]);
                } else if (p.act2 == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = (float)v[e] + (float)rv[e];
                        v[e] = round_to_half(t > 0.f ? t : 0.f);
                    }
                } else {
#pragma nounroll
                    for (int e = 0; e < 8; ++e) v[e] = round_to_half(act_apply((float)v[e] + (float)rv[e], p.act2, p.alpha2));
                }
            }
            if (p.out_i8) {  // requantise: round to nearest even, clamp to +-127, 8 channels = one 8-byte store
                unsigned long long qv = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = rintf((float)v[e] * p.out_inv_scale);
                    t = t > 127.f ? 127.f : (t < -127.f ? -127.f : t);
                    qv |= (unsigned long long)(unsigned char)(int8_t)(int)t << (8 * e);
                }
                *reinterpret_cast<unsigned long long*>(static_cast<int8_t*>(p.out) + (size_t)m * p.ld_out + co) = qv;
            } else if (vec) {
                *reinterpret_cast<half8*>(out + (size_t)m * p.ld_out + co) = v;
            } else {  // ragged channel counts / unaligned slices: element-wise stores
#pragma nounroll
                for (int e = 0; e < 8; ++e)
                    if (co + e < p.Cout) out[(size_t)m * p.ld_out + co + e] = v[e];
            }
        }
    }
}

// ---- epilogue of the fp32 kernels (conv_igemm_f32.hip).  The accumulator fragment of a lane is already what an NHWC fp32 store wants: four consecutive
// channels (16j + 4 (lane >> 4) + [0, 4)) of one pixel = 16 bytes, the four lane groups of a pixel cover 64 contiguous bytes.  No staging through LDS, no
// barrier: activation / shortcut / store straight from the registers (the bias is where the sums started).  SiLU / sigmoid are v * rcp(1 + exp2(-v log2 e)) on the hardware's v_exp_f32 /
// v_rcp_f32 (1 ulp each): 5 instructions per element where expf + an IEEE division are ~30.  Measured on the first build (round 5): with the accurate
// forms a 128 x 80 tile's epilogue was ~7k cycles of VALU issue per wave - 12 % of a 45-step 3x3 and MORE than the whole k-loop of a 4-step 1x1 - for an
// error 60x below what 63 layers of fp32 summation leave on a logit (7e-5 at 640 x 640, against BASELINE's 1e-4).  The row slabs are walked by a rolled
// loop: one copy of the activation code per column fragment, not per accumulator.
__device__ __attribute__((noinline)) float act_slow_f32(float v, int act, float alpha) {
    switch (act) {
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        case ACT_MISH: return mish_ref(v);
        default: return v;
    }
}
__device__ __forceinline__ float act_f32(float v, int act, float alpha) {
    if (act == ACT_NONE) return v;
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU || act == ACT_SIGMOID) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
        return act == ACT_SILU ? v * sg : sg;
    }
    return act_slow_f32(v, act, alpha);
}
template <int NFRAG, int MI, typename PixelOf>
__device__ __forceinline__ void conv_epilogue_f32(const ConvArgs& p, floatx4 (&acc)[MI][NFRAG], int lane, int n0, PixelOf&& pixel_of, int rbase) {
    float* __restrict__ out = static_cast<float*>(p.out);
    const float* __restrict__ res = static_cast<const float*>(p.residual);
    const int px_in = lane & 15;
    const int ch_in = (lane >> 4) * 4;
    const bool vec = !p.scalar_out;   // (the bias is already in the accumulators: conv_igemm_tile starts the sums at it)
#pragma nounroll
    for (int i = 0; i < MI; ++i) {
        floatx4 v[NFRAG];
#pragma unroll
        for (int ii = 0; ii < MI; ++ii) {
            if (ii != i) continue;
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) v[j] = acc[ii][j];
        }
        const int m = pixel_of(rbase + i * 16 + px_in);
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const int co = n0 + j * 16 + ch_in;
            if (m < 0 || co >= p.Cout) continue;
            floatx4 x = v[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = act_f32(x[e], p.act1, p.alpha1);
            if (res) {
                floatx4 r = floatx4{0.f, 0.f, 0.f, 0.f};
                if (vec) {
                    r = *reinterpret_cast<const floatx4*>(res + (size_t)m * p.ld_res + co);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Cout) r[e] = res[(size_t)m * p.ld_res + co + e];
                }
                x += r;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = act_f32(x[e], p.act2, p.alpha2);
            if (vec) {
                *reinterpret_cast<floatx4*>(out + (size_t)m * p.ld_out + co) = x;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < p.Cout) out[(size_t)m * p.ld_out + co + e] = x[e];
            }
        }
    }
}

// TPS = filter taps per k-step: 1 normally; 2 for Cin <= 16 (CinK = 16), where one 32-wide step covers taps 2kt and 2kt+1
// and the tap a lane fetches depends on which half of the row it fills.
// I8: int8 activations / weights on v_mfma_i32_16x16x64_i8 (kINT8 engines).  A 64-byte LDS row then holds 64 int8 channels
// instead of 32 halfs; the host passes the input-side geometry in 2-byte units (see ConvArgs), so the whole operand path
// below is byte-for-byte the fp16 one - only the MFMA and the epilogue's dequantise / requantise differ.
// MI = 16-row MFMA fragments per wave along M: 2 -> a 128-row tile (the default), 1 -> a 64-row tile (twice the workgroups for
// layers whose 128-row tiling leaves most of the 256 CUs idle), 4 -> a 256-row tile (half the tiles, prologues and weight
// traffic for the large maps); chosen per layer by the tactic tuner, runtime/tune.cpp.
// WN = waves along N: 1 -> the four waves stack along M and each reads ALL of the B tile (fine up to 128 x 128: 40 LDS bytes per
// wave-cycle at most); 2 -> a 2 x 2 wave grid, each wave owns (BM / 2) x (BN / 2).  The large-GEMM configuration is <NFRAG 8, BKT 64,
// MI 8, WN 2, three stages>: a 256 x 128 tile of 128 x 64 wave tiles - 12 ds_read_b128 per 32 MFMAs (24 LDS bytes per wave-cycle, 75 %
// of what the CU's LDS delivers when all four SIMDs keep their MFMA pipe full; the 128 x 128 / WN 1 tile needs 160 B/clk for that and
// cannot), 43 MACs per operand byte fetched into LDS, accumulators 128 VGPRs.  Its three stages are 144 KB of LDS: one workgroup per
// CU, which the 64 independent MFMAs per k-step of every wave tolerate.  NSTO overrides the stage count.
// RS = register-staged operands: global memory -> VGPRs (buffer_load_dwordx4) -> LDS (ds_write_b128, the same lane-linear rows the
// DMA writes) instead of buffer_load ... lds.  One LDS-DMA piece (1 KiB per wave-instruction) costs the issuing wave 60-185 cycles
// of issue (MI355X_MICROARCH.md; ablation of this kernel, profiles/r03_gemm_ablation.txt: the k-loop with the MFMAs removed is 72 % of
// the whole kernel and does not shrink when the loads are range-checked away), against 16 for an MFMA: a 128 x 128 x 64 step is 8
// pieces = ~800 cycles next to 512 cycles of MFMA per wave - the loop is bound by DMA issue.  A register load issues in a few cycles;
// the tile for step kt+1 is fetched into registers before the MFMAs of step kt are issued and written to the other LDS stage after them
// (two LDS stages, one barrier per step).  Same LDS contents, same MFMA order: bit-identical results.
// UP = folded nearest 2x upsample (ConvArgs::up_in, 1x1 stride-1 layers): k-steps whose channel offset lies below up_C fetch their A rows
// from the half-resolution tensor at (h >> 1, w >> 1) through a second buffer descriptor; everything after the fetch is unchanged.
// ONE = 1x1 stride-1 unpadded convolution over un-padded channels (Cin == CinK): a plain GEMM.  There is no tap to validate and no ragged
// channel chunk, so the per-step source address of a piece is its pixel's base plus a running byte offset - one VALU add where the general
// walk spends a shift, a mask, a compare and a select per piece (the loop is issue-bound: 2.4 VALU per MFMA, r03_sq_counters_res5_3x3.txt).
// Rows beyond M carry an out-of-range base from the start; the run-out steps past K read whatever follows (they are never multiplied).
// Measured (round 3, same box): the 1x1 layers of YOLOv8n b32 3-10 % faster one at a time (34-layer sum 715 vs 724 us), res5's 1x1
// 2048 -> 512 GEMM 497 vs 509 us; bench.py within run-to-run noise (35.0-35.2k vs 34.4-35.3k img/s).  Same bits.
// LDS bytes of one instantiation (the stage buffers, or the epilogue's staging tiles if those are larger)
template <int NFRAG, int BKT, int MI, int WN, int NSTO, int NW, bool RS>
constexpr int igemm_lds_bytes() {
    constexpr int BN = 16 * NFRAG, WM = NW / WN, NFW = NFRAG / WN, BM = WM * 16 * MI, ROW_B = BKT * 2, CH = BKT / 8, RPI = 64 / CH;
    constexpr int B_PASSES = (BN + NW * RPI - 1) / (NW * RPI), B_ROWS = B_PASSES * NW * RPI;
    constexpr int STAGE_BYTES = BM * ROW_B + B_ROWS * ROW_B;
    constexpr int NST = RS ? 2 : NSTO ? NSTO : (BKT == 64 ? 2 : 3);
    constexpr int EPI_BYTES = NW * 16 * (16 * NFW * 4 + 16);
    return NST * STAGE_BYTES > EPI_BYTES ? NST * STAGE_BYTES : EPI_BYTES;
}

// One (BM x BN) output tile at (m0, n0) of the convolution `p`: the whole kernel but for the blockIdx -> tile mapping, which the two
// entry points below do differently (one problem per launch / several problems per launch, round 4).
// ROLES (round 5): the workgroup is 2 * NW waves; waves NW .. 2NW-1 only FETCH (address arithmetic, the LDS-DMA pieces, the counted wait, the barrier) and
// waves 0 .. NW-1 only MULTIPLY (barrier, fragment reads, MFMAs, epilogue).  A DMA piece holds its wave's issue port for 60-200 cycles and the tap / address
// arithmetic in front of it for a few dozen more; in the one-role kernel that is time the same wave's MFMAs are not being issued (a wave issues in order), and
// with 3-4 waves per SIMD the others cover it only partly: tools/hip/mfma_f32_rate.hip measures the bare skeleton at 0.66 / 0.77 / 0.82 of the fp32 MFMA peak
// with 1 / 2 / 4 workgroups per CU, and the same skeleton with roles at 0.80 / 0.89 / 0.89 (profiles/r05_mfma_f32_rate_roles.txt).  Same tiles, same LDS
// contents, same K order: the same bits as the one-role kernel.  Instantiated for the fp32 tiles (conv_igemm_f32.hip, tactic ws == 6: the tuner takes it on the
// 80-wide detect-head layers, -8..-16 % there).  On the fp16 128-row tiles it was measured too: -3..-17 % on small maps alone, nothing on the three-context
// bench line (profiles/r05_fp16_roles_*) - not instantiated there.
template <int NFRAG, int BKT, int TPS, bool I8 = false, int MI = 2, int WN = 1, int NSTO = 0, bool PRE = false, int NW = 4, bool RS = false, bool UP = false,
          bool ONE = false, bool F32 = false, bool ROLES = false>
__device__ __forceinline__ void conv_igemm_tile(const ConvArgs& p, unsigned in_bytes, unsigned w_bytes, const int m0, const int n0, int dbg_flags,
                                                char* __restrict__ smem) {
    const int dbg = TRTX_DBG(dbg_flags);
    TRTX_MARK(0);
    constexpr int BN = 16 * NFRAG;
    constexpr int WM = NW / WN;                    // waves along M (NW = waves per workgroup: 4, or 8 for the large-GEMM tile)
    constexpr int NFW = NFRAG / WN;                // 16-column fragments per wave
    static_assert(NFRAG % WN == 0 && (WN == 1 || WN == 2), "wave grid");
    constexpr int BM = WM * 16 * MI;               // rows per tile: every wave owns 16 * MI of them
    constexpr int WR = 16 * MI;                    // rows per wave
    constexpr int ROW_B = BKT * 2;                 // bytes per LDS row
    constexpr int CH = BKT / 8;                    // 16-byte chunks per row
    constexpr int RPI = 64 / CH;                   // rows filled by one wave-instruction (16 / 8)
    constexpr int A_LOADS = BM / (NW * RPI);       // per wave per k-step (2 / 4)
    constexpr int B_PASSES = (BN + NW * RPI - 1) / (NW * RPI);
    constexpr int B_ROWS = B_PASSES * NW * RPI;    // rows beyond BN are dummy targets
    constexpr int A_BYTES = BM * ROW_B;
    constexpr int STAGE_BYTES = A_BYTES + B_ROWS * ROW_B;
    constexpr int LOADS_PER_TILE = A_LOADS + B_PASSES;
    constexpr int KSUB = BKT / 32;                 // MFMA k-slices per k-step
    constexpr int NST = RS ? 2 : NSTO ? NSTO : (BKT == 64 ? 2 : 3);  // pipeline stages (64-wide stages are double-buffered to keep occupancy)
    constexpr int EPI_BYTES = NW * 16 * (16 * NFW * 4 + 16);   // the epilogue's wave-private staging tiles (conv_epilogue)
    constexpr int LDS_BYTES = NST * STAGE_BYTES > EPI_BYTES ? NST * STAGE_BYTES : EPI_BYTES;
    static_assert(LDS_BYTES == igemm_lds_bytes<NFRAG, BKT, MI, WN, NSTO, NW, RS>(), "the entry points allocate what the tile function uses");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool fetcher = ROLES && wave_all >= NW;            // wave-uniform
    const int wave = ROLES ? (wave_all & (NW - 1)) : wave_all;   // a fetching wave fills the LDS rows the one-role kernel's wave of the same index fills
    static_assert(!ROLES || (!RS && NW == 4), "roles: LDS-DMA operands, 4 + 4 waves");

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, w_bytes, 0x00020000);
    const unsigned up_bytes = UP ? (unsigned)((((size_t)p.N * p.up_H * p.up_W - 1) * p.up_ld + p.up_C) * 2) : 0u;
    const __amdgpu_buffer_rsrc_t rs_up = UP ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.up_in), 0, up_bytes, 0x00020000) : rs_in;

    // ---- per-lane source description.  Instruction i of wave w fills LDS rows (4i + w) * RPI + [0, RPI); lane l
    // writes row + l / CH, physical chunk l % CH, so it fetches LOGICAL chunk (l % CH) ^ swz(row).
    const int lrow = lane / CH;
    const int lswz = BKT == 32 ? swz<32>(lrow) : ((lrow >> 1) | ((wave & 1) << 2));  // row base is a multiple of RPI
    const int lchunk = (lane % CH) ^ lswz;
    const int tsel = TPS == 2 ? (lchunk >> 1) : 0;         // which of the step's taps this lane fetches
    const int cchunk = TPS == 2 ? (lchunk & 1) : lchunk;   // 8-channel chunk inside the tap
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = __builtin_amdgcn_rcpf((float)HoWo), inv_wo = __builtin_amdgcn_rcpf((float)p.Wo);  // +-1 estimates, fixed up
    unsigned a_base[A_LOADS];   // byte offset of (n, hi0, wi0, channel lchunk*8); wraps for border pixels (masked)
    unsigned a_rows[A_LOADS];   // bit r: filter row r of this pixel lies inside the image (0 for pixels >= M)
    unsigned a_cols[A_LOADS];   // bit q: filter column q lies inside the image
    unsigned a_up[UP ? A_LOADS : 1];   // UP: byte offset of (n, ho >> 1, wo >> 1, channel lchunk*8) in the half-resolution tensor
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + (NW * i + wave) * RPI + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        // quotients are small (image index, output row): a float estimate is within +-1, fixed up exactly
        int n = (int)((float)mm * inv_howo);
        int rem = mm - n * HoWo;
        if (rem < 0) { --n; rem += HoWo; }
        if (rem >= HoWo) { ++n; rem -= HoWo; }
        int ho = (int)((float)rem * inv_wo);
        int wo = rem - ho * p.Wo;
        if (wo < 0) { --ho; wo += p.Wo; }
        if (wo >= p.Wo) { ++ho; wo -= p.Wo; }
        const int hi0 = ho * p.stride_h - p.pad_h;
        const int wi0 = wo * p.stride_w - p.pad_w;
        a_base[i] = (unsigned)(((n * p.H + hi0) * p.W + wi0) * p.ld_in + cchunk * 8) * 2u;  // element index < 2^30 (slice < 2 GB)
        if constexpr (UP) a_up[i] = (unsigned)(((n * p.up_H + (ho >> 1)) * p.up_W + (wo >> 1)) * p.up_ld + cchunk * 8) * 2u;
        if constexpr (ONE) a_base[i] = ok ? a_base[i] : kOOB;   // (kOOB + any k offset < 2^30 stays out of range)
        // taps inside the image form a contiguous range (dilation 1): closed form instead of a loop over taps
        a_rows[i] = ok ? tap_range_mask(hi0, p.kh, p.H) : 0u;
        a_cols[i] = tap_range_mask(wi0, p.kw, p.W);
    }
    const int cmax = p.Cin - cchunk * 8;  // this lane's chunk holds real channels while uc < cmax
    unsigned b_off[B_PASSES];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
        const int row = (NW * j + wave) * RPI + lrow;
        b_off[j] = row < BN ? (unsigned)(((n0 + row) * p.Kpad + lchunk * 8) * 2) : kOOB;  // weights < 2 GB
    }
    // Column tiles that are not a whole number of DMA passes (BN = 16 / 32 / 80 with 32-wide steps, 80 with 64-wide ones): in the LAST pass
    // some waves' pieces lie wholly beyond the tile - rows nobody reads.  Those waves skip the instruction (round 4: a DMA piece costs its
    // wave 60-185 cycles of issue and the loop is bound by exactly that; on the 80-wide detect-head arms three of four waves issued a
    // dead piece per k-step) and wait on one load fewer per tile in flight.  Same LDS contents wherever anything is read: same bits.
    constexpr bool B_PARTIAL = !RS && (BN % (NW * RPI) != 0);
    const bool b_last_live = !B_PARTIAL || (NW * (B_PASSES - 1) + wave) * RPI < BN;   // wave-uniform

    // wave-uniform walk over K.  TPS == 1: the (tap, channel offset, byte offset) of k-step e is precomputed by lane e & 63 into
    // two VGPRs (a 64-step window, rebuilt every 64 steps) and fetched with v_readlane: the kernel is instruction-issue bound
    // and this replaces ~25 scalar instructions per step.  Validity of (pixel, tap) is one bit of a per-pixel tap mask (rows x
    // cols expanded once, below).  TPS == 2 keeps (r, q) of the two taps of a step and every lane selects its own.
    const int nk = p.Kpad / BKT;
    int s_kt = 0, s_uc = 0;
    unsigned a_taps[A_LOADS];
    unsigned t_add = 0, t_tap = 0;  // lane l: byte offset / (tap | uc << 8) of k-step window_base + l
    const int spt = p.CinK / BKT;   // k-steps per tap (TPS == 1)
    const float inv_spt = __builtin_amdgcn_rcpf((float)spt), inv_kw = __builtin_amdgcn_rcpf((float)p.kw);
    auto build_window = [&](int base) {
        const int e = base + lane;
        const int tap = (int)(((float)e + 0.5f) * inv_spt);  // (e + 0.5) / spt is >= 1/128 away from an integer: 1-ulp rcp is exact enough
        const int ucs = (e - tap * spt) * BKT;
        const int r = (int)(((float)tap + 0.5f) * inv_kw);
        const int q = tap - r * p.kw;
        t_add = (unsigned)((r * p.dil_h * p.W + q * p.dil_w) * p.ld_in + ucs) * 2u;
        t_tap = (unsigned)(tap < 31 ? tap : 31) | ((unsigned)ucs << 8);  // tap masks have no bit 31: run-out steps are dead
    };
    if (TPS == 1) {
        build_window(0);
        // tap (r, q) is bit r*kw + q: the column mask replicated into every filter row (one multiply by the constant
        // sum_r 2^(r*kw)), restricted to the contiguous range of valid rows
        unsigned rep = 0;
        for (int r = 0; r < p.kh; ++r) rep |= 1u << (r * p.kw);  // wave-uniform (scalar)
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const unsigned rows = a_rows[i];
            const int lo = rows ? __builtin_ctz(rows) : 0, hi = rows ? 32 - __builtin_clz(rows) : 0;  // valid rows [lo, hi)
            const unsigned range = hi > lo ? ((hi * p.kw >= 32 ? ~0u : ((1u << (hi * p.kw)) - 1u)) & ~((1u << (lo * p.kw)) - 1u)) : 0u;
            a_taps[i] = (a_cols[i] * rep) & range;
        }
    }
    const bool full_c = p.Cin == p.CinK;  // no ragged channel chunk to mask
    int s_r[TPS], s_q[TPS];
    unsigned s_toff[TPS];
    auto tap_next = [&](int& r, int& q) {  // select arithmetic, no branches
        ++q;
        const int wq = q == p.kw;
        q = wq ? 0 : q;
        r += wq;
    };
    s_r[0] = 0;
    s_q[0] = 0;
    if constexpr (TPS == 2) {
        s_r[1] = 0;
        s_q[1] = 0;
        tap_next(s_r[1], s_q[1]);
    }
#pragma unroll
    for (int t = 0; t < TPS; ++t) s_toff[t] = (unsigned)((s_r[t] * p.dil_h * p.W + s_q[t] * p.dil_w) * p.ld_in) * 2u;

    intx4 ra[RS ? A_LOADS : 1], rb[RS ? B_PASSES : 1];   // RS: the operand tile in flight (registers)
    auto issue_tile = [&](int stage) {
        char* sbase = smem + stage * STAGE_BYTES;
        const bool live = s_kt < nk && !(dbg & 1);
        if constexpr (ONE) {
            const unsigned koff = (unsigned)s_kt * (unsigned)(BKT * 2);   // scalar
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const unsigned voff = a_base[i] + koff;
                if constexpr (RS) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sbase + (NW * i + wave) * RPI * ROW_B), 16, voff, 0, 0, 0);
            }
        } else if (TPS == 1) {
            const unsigned add = (unsigned)__builtin_amdgcn_readlane((int)t_add, s_kt & 63);
            const unsigned tw = (unsigned)__builtin_amdgcn_readlane((int)t_tap, s_kt & 63);
            const int tap = (int)(tw & 255u), uc = (int)(tw >> 8);
            const bool chunk_ok = full_c || uc < cmax;
            const bool from_up = UP && uc < p.up_C;          // wave-uniform: this k-step's channels live in the half-resolution tensor
            const __amdgpu_buffer_rsrc_t rs_a = from_up ? rs_up : rs_in;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const bool ok = ((a_taps[i] >> tap) & 1u) && chunk_ok && live;
                unsigned voff = ok ? a_base[i] + add : kOOB;
                if constexpr (UP) voff = from_up ? (ok ? a_up[i] + (unsigned)uc * 2u : kOOB) : voff;   // 1x1: `add` is the channel offset alone
                if constexpr (RS) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(sbase + (NW * i + wave) * RPI * ROW_B), 16, voff, 0, 0, 0);
            }
        } else {
            const int r = tsel ? s_r[TPS - 1] : s_r[0];
            const int q = tsel ? s_q[TPS - 1] : s_q[0];
            const unsigned add = (tsel ? s_toff[TPS - 1] : s_toff[0]) + (unsigned)s_uc * 2u;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const bool ok = ((a_rows[i] >> r) & (a_cols[i] >> q) & 1u) && s_uc < cmax && live;
                const unsigned voff = ok ? a_base[i] + add : kOOB;
                if constexpr (RS) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sbase + (NW * i + wave) * RPI * ROW_B), 16, voff, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) {
            if (B_PARTIAL && j == B_PASSES - 1 && !b_last_live) continue;   // this wave's piece of the last pass lies beyond the column tile
            const unsigned voff = ONE ? b_off[j] : ((s_kt < nk && !(dbg & 2)) ? b_off[j] : kOOB);
            if constexpr (RS) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sbase + A_BYTES + (NW * j + wave) * RPI * ROW_B), 16, voff, 0, 0, 0);
            b_off[j] += BKT * 2;  // kOOB stays out of range for any K < 2^30
        }
        ++s_kt;
        if (TPS == 2) {
#pragma unroll
            for (int t = 0; t < TPS; ++t) {
                tap_next(s_r[t], s_q[t]);
                tap_next(s_r[t], s_q[t]);
                s_toff[t] = (unsigned)((s_r[t] * p.dil_h * p.W + s_q[t] * p.dil_w) * p.ld_in) * 2u;
            }
        } else if (!ONE && (s_kt & 63) == 0) {
            build_window(s_kt);  // next 64-step window (uniform, once per 64 steps)
        }
    };

    // RS: the fetched tile -> LDS stage `stage`, lane-linear rows exactly where the DMA would have put them
    auto commit = [&](int stage) {
        char* sbase = smem + stage * STAGE_BYTES + lane * 16;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) *reinterpret_cast<intx4*>(sbase + (NW * i + wave) * RPI * ROW_B) = ra[i];
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<intx4*>(sbase + A_BYTES + (NW * j + wave) * RPI * ROW_B) = rb[j];
    };

    floatx4 acc[MI][NFW];
    intx4 acci[MI][NFW];
    floatx4 part[F32 ? MI : 1][F32 ? NFW : 1];   // fp32: the running k-step's partial sum (two-level sum, see compute())
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            acci[i][j] = intx4{0, 0, 0, 0};
            if constexpr (F32) part[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    if constexpr (F32) {
        // fp32: the running total STARTS at the bias (a lane's fragment is four consecutive channels of one pixel, the same four for every row slab), so
        // the fetch passes under the first tiles' round trip to memory instead of standing, a full L2 latency, in front of the epilogue's first store
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < NFW; ++j) {
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(p.bias + n0 + (WN == 1 ? 0 : (wave & 1)) * NFW * 16 + j * 16 + (lane >> 4) * 4);   // [Cout_pad] values
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = b4;
            }
        }
    }

    // fragment read offsets inside a stage: row (lane & 15), logical chunk (lane >> 4) [+ 4 for the second k-slice]
    const int frow = lane & 15;
    const int fswz = swz<BKT>(frow);
    int f_off[KSUB];
#pragma unroll
    for (int h = 0; h < KSUB; ++h) f_off[h] = frow * ROW_B + ((((lane >> 4) + 4 * h) ^ fswz) * 16);
    const int wave_m = WN == 1 ? wave : (wave >> 1), wave_n = WN == 1 ? 0 : (wave & 1);
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    const int a_frag = wave_m * WR * ROW_B;
    const int b_frag = A_BYTES + wave_n * NFW * 16 * ROW_B;

    auto compute = [&](int stage) {
        const char* sb = smem + stage * STAGE_BYTES;
        if constexpr (F32) {
            // fp32 operands (conv_igemm_f32.hip): a 64-byte LDS row is 16 floats, a lane's 16-byte fragment read is floats 4g .. 4g+3 (g = lane >> 4) of
            // its row, and component s of it is the lane's operand of the s-th v_mfma_f32_16x16x4_f32 - that instruction's k index (lane >> 4) then
            // stands for channel 4g + s on BOTH operands, so four of them sum the 16 channels of the step.  Every fragment is read before the first
            // MFMA and consecutive MFMAs name different accumulators (the instruction issues every 32 cycles, its result is back after 40).
            // TWO-LEVEL SUM.  The MFMA is an fmaf chain: accumulating all of K in one register is a K-long sequential sum, whose round-off grows like
            // sqrt(K) ulp of the running total (K = 576: ~24) - measured on the first build, 1.5e-4 on YOLOv8n's head logits at 640 x 640 against
            // BASELINE's 1e-4, twice what the blocked sums of a CPU library leave.  So a k-step's 16 products are summed from zero (srcC = 0) and the
            // step's partial is then added to the running total by the vector ALU: chains of 16 and K / 16 instead of one of K (K = 576: ~4 + 6 ulp).
            // Four v_add_f32 per fragment and step in the shadow of 128 cycles of MFMA; costs a second set of accumulator registers.
            // The partial of step kt is added while step kt+1 runs: each fragment's add stands right in front of the MFMA that restarts its partial from
            // zero, in the shadow of the MFMA issued before it (adds at the END of a step would wait for the step's last MFMAs and hold the wave's issue
            // port for ~32 VALU instructions with the matrix pipe idle: measured -6 %).  `part` lives across steps; the last one is added after the loop.
            // BKT == 64: a 32-channel step = 128-byte rows, whole cache lines per row (half the line requests per byte of the fill path - what bounds this
            // kernel, DESIGN 5 "round 5"); computed as its two 16-channel halves one after the other, each with its own partial: the arithmetic of two
            // 16-channel steps, so both step widths return the same bits.
            static_assert(!F32 || (!I8 && !PRE), "fp32 operands");
            __builtin_amdgcn_s_setprio(1);   // a wave in its MFMA phase goes first: the matrix pipe is what this kernel is priced on, and co-resident waves in
                                             // their set-up / epilogue (VALU, transcendentals on the same issue port) otherwise open bubbles in it
#pragma unroll
            for (int h = 0; h < KSUB; ++h) {
                floatx4 af[MI], bf[NFW];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const floatx4*>(sb + a_frag + i * 16 * ROW_B + f_off[h]);
#pragma unroll
                for (int j = 0; j < NFW; ++j) bf[j] = *reinterpret_cast<const floatx4*>(sb + b_frag + j * 16 * ROW_B + f_off[h]);
#pragma unroll
                for (int j = 0; j < NFW; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        acc[i][j] += part[i][j];
                        part[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][0], af[i][0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    }
#pragma unroll
                for (int s4 = 1; s4 < 4; ++s4)
#pragma unroll
                    for (int j = 0; j < NFW; ++j)
#pragma unroll
                        for (int i = 0; i < MI; ++i) part[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s4], af[i][s4], part[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            return;
        }
#pragma unroll
        for (int h = 0; h < KSUB; ++h) {
            if constexpr (I8) {
                intx4 af[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const intx4*>(sb + a_frag + i * 16 * ROW_B + f_off[h]);
#pragma unroll
                for (int j = 0; j < NFW; ++j) {
                    const intx4 bf = *reinterpret_cast<const intx4*>(sb + b_frag + j * 16 * ROW_B + f_off[h]);
#pragma unroll
                    for (int i = 0; i < MI; ++i) acci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bf, af[i], acci[i][j], 0, 0, 0);
                }
            } else if constexpr (!PRE) {
                half8 af[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const half8*>(sb + a_frag + i * 16 * ROW_B + f_off[h]);
#pragma unroll
                for (int j = 0; j < NFW; ++j) {
                    const half8 bf = *reinterpret_cast<const half8*>(sb + b_frag + j * 16 * ROW_B + f_off[h]);
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, af[i], acc[i][j], 0, 0, 0);
                }
            }
        }
        if constexpr (PRE && !I8) {
            // every fragment read of the k-step is issued before its first MFMA (B fragment 0 first, then the A fragments, then the
            // rest): the LDS returns in order, so the first MFMA waits for two reads and the later ones find theirs landed - with one
            // workgroup per CU there is no other wave to cover a read issued right before its use
            half8 af[KSUB][MI], bf[KSUB][NFW];
#pragma unroll
            for (int h = 0; h < KSUB; ++h) {
                bf[h][0] = *reinterpret_cast<const half8*>(sb + b_frag + f_off[h]);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[h][i] = *reinterpret_cast<const half8*>(sb + a_frag + i * 16 * ROW_B + f_off[h]);
#pragma unroll
                for (int j = 1; j < NFW; ++j) bf[h][j] = *reinterpret_cast<const half8*>(sb + b_frag + j * 16 * ROW_B + f_off[h]);
            }
            __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks every read next to its MFMA again (two live B fragments)
#pragma unroll
            for (int h = 0; h < KSUB; ++h)
#pragma unroll
                for (int j = 0; j < NFW; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[h][j], af[h][i], acc[i][j], 0, 0, 0);
        }
    };

    // one k-step: tile kt has landed once at most the LOADS_PER_TILE loads of tile kt+1 are still in flight; the
    // barrier also guarantees every wave finished reading the stage that is refilled next - FINISHED, not just issued: the wait names
    // lgkmcnt(0) too.  Nothing else makes the compiler complete the previous step's fragment reads before this barrier (s_barrier is not a
    // fence; the asm's "memory" clobber orders the issue of memory operations, not their completion, and the MFMAs that consume the
    // fragments may be scheduled below the barrier).  In the row-reuse kernel's three-stage instantiations it did sink 1-7 reads and their
    // MFMAs below the barrier, and under co-scheduling the next tile's range-checked-away DMA pieces (zero fill, no memory round trip)
    // overtook them: the hazard of round 3 (profiles/r04_r3_bisect.txt).  In this kernel's instantiations the reads were complete anyway
    // (tools/isa_barrier_reads.py over the compiled ISA, tests/test_isa_barrier_reads.py); now the source says so.
#define TRTX_KSTEP(S)                                                         \
    {                                                                         \
        TRTX_STAMP(0, kt);                                                    \
        if (B_PARTIAL && !b_last_live) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((LOADS_PER_TILE - 1) * (NST - 2)) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS_PER_TILE * (NST - 2)) : "memory"); \
        TRTX_STAMP(1, kt);                                                    \
        __builtin_amdgcn_s_barrier();                                         \
        TRTX_STAMP(2, kt);                                                    \
        issue_tile(((S) + NST - 1) % NST);                                    \
        TRTX_STAMP(3, kt);                                                    \
        if (!(dbg & 4)) compute(S);                                           \
        TRTX_STAMP(4, kt);                                                    \
    }

    if constexpr (ROLES) {
        if (fetcher) {
#pragma unroll
            for (int st = 0; st < NST - 1; ++st) issue_tile(st);
            for (int kt = 0;;) {
                bool done = false;
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    if (done) continue;
                    // tile kt has landed (at most the NST - 2 younger tiles in flight) - said to the multiplying waves by the barrier
                    if (B_PARTIAL && !b_last_live) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LOADS_PER_TILE - 1) * (NST - 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_TILE * (NST - 2)) : "memory");
                    __builtin_amdgcn_s_barrier();
                    issue_tile((st + NST - 1) % NST);   // into the stage the multiplying waves finished with before this barrier
                    if (++kt == nk) done = true;
                }
                if (done) break;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-out tiles' LDS writes (range-checked away) retire before the wave ends
            if constexpr (!F32) __builtin_amdgcn_s_barrier();  // ... and before the multiplying waves' epilogue reuses the stages as staging tiles (below)
            return;                                             // (an ended wave no longer counts at the workgroup's barriers)
        }
        for (int kt = 0;;) {
            bool done = false;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (done) continue;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous step's fragment reads are COMPLETE before the stage is given back
                __builtin_amdgcn_s_barrier();
                compute(st);
                if (++kt == nk) done = true;
            }
            if (done) break;
        }
        if constexpr (!F32) __builtin_amdgcn_s_barrier();   // the fetching waves' last LDS writes have retired (the fp32 epilogue does not touch LDS)
    } else if constexpr (RS) {
        // registers <- tile 0; LDS stage 0 <- registers; registers <- tile 1.  Step kt: (writes of tile kt visible) barrier, MFMAs of
        // stage kt & 1, then tile kt+1 goes from the registers to the other stage - free since every wave passed this step's barrier
        // after its reads of step kt-1 - and tile kt+2 is fetched.  The compiler places the vmcnt / lgkmcnt waits of the register path.
        issue_tile(0);
        commit(0);
        issue_tile(0);
        for (int kt = 0; !(dbg & 16);) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!(dbg & 4)) compute(kt & 1);
            commit((kt + 1) & 1);
            issue_tile(0);
            if (++kt == nk) break;
        }
    } else {
    // NST - 1 tiles in flight (two normally; the deep variants of the short, latency-bound layers keep five), then one k-step per stage in turn
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) {
        issue_tile(st);
        if (st == 0) TRTX_MARK(1);
    }
    for (int kt = 0; !(dbg & 16);) {
        bool done = false;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            if (done) continue;
            TRTX_KSTEP(st);
            if (++kt == nk) done = true;
        }
        if (done) break;
    }
    }
#undef TRTX_KSTEP
    // the two run-out tiles were range-checked away (no memory access) but their LDS writes must retire before exit
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TRTX_MARK(2);

    if (dbg & 8) return;
    if constexpr (F32) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NFW; ++j) acc[i][j] += part[i][j];   // the last step's partial
        conv_epilogue_f32<NFW, MI>(p, acc, lane, n0 + wave_n * NFW * 16, [&](int t) {
            const int m = m0 + t;
            return m < p.M ? m : -1;
        }, wave_m * WR);
    } else {
    conv_epilogue<NFW, MI, I8, LDS_BYTES, NW>(p, acc, acci, smem, wave, lane, n0 + wave_n * NFW * 16, [&](int t) {
        const int m = m0 + t;
        return m < p.M ? m : -1;
    }, wave_m * WR);
    }
    TRTX_MARK(3);
}

template <int NFRAG, int BKT, int TPS, bool I8 = false, int MI = 2, int WN = 1, int NSTO = 0, bool PRE = false, int NW = 4, bool RS = false, bool UP = false,
          bool ONE = false, bool F32 = false, bool ROLES = false>
__global__ __launch_bounds__(NW * 64 * (ROLES ? 2 : 1)) void conv_igemm_f16_kernel(const ConvArgs p, unsigned in_bytes, unsigned w_bytes, int tiles_n,
                                                             int total_tiles, int xcd_chunk, int dbg_flags) {
    __shared__ __attribute__((aligned(16))) char smem[igemm_lds_bytes<NFRAG, BKT, MI, WN, NSTO, NW, RS>()];   // up to 144 KB of the CU's 160 KB
    int tile = blockIdx.x;
    if (xcd_chunk) {  // XCD-aware order: id -> (xcd = id % 8, slot = id / 8) -> contiguous tile range per XCD
        tile = (tile & 7) * xcd_chunk + (tile >> 3);
        if (tile >= total_tiles) return;
    }
    const int m0 = (tile / tiles_n) * ((NW / WN) * 16 * MI);
    const int n0 = (tile % tiles_n) * (16 * NFRAG);
    conv_igemm_tile<NFRAG, BKT, TPS, I8, MI, WN, NSTO, PRE, NW, RS, UP, ONE, F32, ROLES>(p, in_bytes, w_bytes, m0, n0, dbg_flags, smem);
}

// Several INDEPENDENT convolutions in one launch (round 4).  The YOLOv8 detect head is six chains of depth three over
// three pyramid levels (yolov8/src/model.cpp:188-251): at batch 32 the 20x20 level is 100 tiles for 256 CUs and the 40x40 level 400, each
// paying the per-launch floor on its own.  Here the tile index runs over the tiles of up to kMaxGroup problems of ONE instantiation
// (same column-tile width, k-step, operand path): a workgroup looks its problem up in a prefix table held in the kernel arguments and then
// is exactly a workgroup of that problem's own launch - same tile, same K order, same bits.  The small levels' tiles fill the CUs the
// large level's tail leaves idle, and 18 launches become 6.
// Tile order (measured, round 4: one XCD-contiguous range over the concatenated tiles gave the last XCD all of the small levels' tiles -
// the ones with the 4x longer K - and the cv2.x.0 group ran 108 us against 78 for its three members one after the other): EVERY problem is
// split into eight XCD chunks of its own (its tiles keep their L2 locality), an XCD walks its chunk of problem 0, then of problem 1, ...,
// and the host orders the problems by falling k-steps per tile, so that the long tiles start first and the short ones fill the tail.
struct ConvGroupArgs {
    int n;
    int slot_start[kMaxConvGroup + 1];   // per XCD: prefix sums of the problems' chunk sizes (slot -> problem)
    int chunk[kMaxConvGroup];            // tiles of problem p per XCD: ceil(tiles[p] / 8)
    int tiles[kMaxConvGroup];
    int tiles_n[kMaxConvGroup];
    unsigned in_bytes[kMaxConvGroup], w_bytes[kMaxConvGroup];
    ConvArgs p[kMaxConvGroup];
};
template <int NFRAG, int BKT, bool RS, bool ONE, bool I8 = false>   // I8 (round 5): int8 operands - the members of an INT8 plan's sibling layers
__global__ __launch_bounds__(256) void conv_igemm_group_f16_kernel(const ConvGroupArgs g, int dbg_flags) {
    __shared__ __attribute__((aligned(16))) char smem[igemm_lds_bytes<NFRAG, BKT, 2, 1, 0, 4, RS>()];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int pid = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvGroup; ++k) pid += (k < g.n && slot >= g.slot_start[k]) ? 1 : 0;
    pid = __builtin_amdgcn_readfirstlane(pid);
    const int local = xcd * g.chunk[pid] + (slot - g.slot_start[pid]);
    if (local >= g.tiles[pid]) return;
    const int tn = g.tiles_n[pid];
    const int m0 = (local / tn) * 128;
    const int n0 = (local % tn) * (16 * NFRAG);
    conv_igemm_tile<NFRAG, BKT, 1, I8, 2, 1, 0, false, 4, RS, false, ONE>(g.p[pid], g.in_bytes[pid], g.w_bytes[pid], m0, n0, dbg_flags, smem);
}

}  // namespace
}  // namespace trtx
