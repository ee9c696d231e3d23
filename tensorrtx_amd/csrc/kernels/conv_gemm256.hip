// Large plain-GEMM convolutions (1x1, stride 1, unpadded) on a 256 x 256 x 64 tile with eight waves in two ROLE-ALTERNATING groups - the
// k-loop skeleton measured in tools/hip/gemm8p.hip (round 4), brought behind conv_tactics() for the layers it was
// measured on: R-CNN's res5 1x1 GEMMs on 4000 RoIs (rcnn/backbone.hpp:100-229: M = 196 000, N = 512 / 2048, K = 512 / 1024 / 2048) ran at
// 0.25-0.33 of the 2.5 PFLOP/s MFMA peak on the 128 x 128 tile; this skeleton measured 0.41-0.45 on K = 2048, 0.33-0.38 on K = 1024,
// 0.29-0.33 on K = 512 (profiles/r04_gemm8p_skeleton.txt).
//
// Structure (MI355X_MICROARCH.md "Two waves per SIMD"):
//   * 8 waves = 2 (M) x 4 (N) wave tiles of 128 x 64; waves w and w + 4 share a SIMD and belong to different role groups (wm = w >> 2);
//   * a K-tile (64) is FOUR PHASES, one C quadrant (64 x 32 per wave: 16 x v_mfma_f32_16x16x32_f16) each; a phase is a LOAD segment
//     (fragment ds_reads of this phase + 2 LDS-DMA pieces of a half-tile up to two K-tiles ahead + the counted wait) and a COMPUTE segment
//     (MFMAs only), separated by s_barrier.  Group 1 runs one barrier behind group 0: on every SIMD one wave multiplies while its partner
//     loads - the MFMA-issuing wave issues no DMA and no ds_read.  The younger group holds a static s_setprio 1;
//   * LDS: 2 buffers x (A 256 x 64 + B 256 x 64) fp16 = 128 KB (one workgroup per CU), each operand in two 128-row half-tiles of 16 KB;
//     16-byte chunks XOR-swizzled on the SOURCE side ((row >> 1) & 7: conflict-free fragment reads);
//   * reads:  P0 a0,b0(E)  P1 b1(E)  P2 a1(E)  P3 -   P4 a0,b0(O)  P5 b1(O)  P6 a1(O)  P7 -        (E / O: K-tiles 2i / 2i + 1, buffers 0 / 1)
//     stages: P0 A-lo(O)   P1 A-hi(O)  P2 B-lo(E+2)  P3 B-hi(E+2)  P4 A-lo(E+2)  P5 A-hi(E+2)  P6 B-lo(O+2)  P7 B-hi(O+2)
//     every region is re-staged at least one full phase after its last ds_read retired; counted waits only at P3 / P7 (vmcnt(4)).
// Every output element accumulates K in ascending 32-wide slices through the same MFMA as conv_igemm_f16_kernel, and the epilogue is that
// kernel's arithmetic (fp32 bias, act1, ONE fp16 rounding, residual add on the rounded value, act2): the same bits, so the tile is one more
// exchangeable tactic (ConvArgs::bn == 256), timed against the others when the plan is built.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../common.h"
#include "kernels.h"
#include "launch.h"

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr unsigned kOOB = 0x80000000u;
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROW_B = BK * 2;            // 128 bytes per LDS row
constexpr int HALF_B = 128 * ROW_B;      // one half-tile: 128 rows = 16 KB
constexpr int BUF_B = 4 * HALF_B;        // A-lo, A-hi, B-lo, B-hi
constexpr int LDS_B = 2 * BUF_B;         // 128 KB

// the activations of conv_igemm.hip's epilogue, expression for expression (bit-identical outputs are the contract)
__device__ __attribute__((noinline)) float g256_act_slow(float v, int act, float alpha) {
    switch (act) {
        case ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        case ACT_MISH: return mish_ref(v);
        default: return v;
    }
}
__device__ __forceinline__ float g256_act(float v, int act, float alpha) {
    if (act == ACT_NONE) return v;
    if (act == ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    return g256_act_slow(v, act, alpha);
}

// CONV3: the same GEMM over a 3x3 stride-1 pad-1 convolution's implicit A matrix (Cin a whole number of 64-channel K-tiles; K runs
// (tap, channel slice) like the packed weights): a lane's four A rows carry the byte offset of their pixel's top-left tap and a 9-bit mask of the
// taps that lie inside the image; stage() adds the K-tile's tap delta or range-checks the piece away (zero fill = the padding).
template <bool CONV3>
__global__ __launch_bounds__(512) void conv_gemm256_f16_kernel(const ConvArgs p, unsigned a_bytes, unsigned w_bytes, int tiles_n, int total_tiles, int xcd_chunk) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_B];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tile = blockIdx.x;
    tile = (tile & 7) * xcd_chunk + (tile >> 3);
    if (tile >= total_tiles) return;
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int nk = p.Kpad / BK;
    const int lda = p.ld_in, ldw = p.Kpad;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, w_bytes, 0x00020000);

    // DMA source offsets: piece i (0, 1) of wave w fills rows (8 i + w) * 8 + [0, 8) of a half-tile; lane l writes row + (l >> 3), physical
    // chunk l & 7, so it fetches logical chunk (l & 7) ^ f(row), f(row) = (row >> 1) & 7 = (l >> 4) | (w & 1) << 2
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ ((lane >> 4) | ((wave & 1) << 2));
    unsigned a_off[2][2], w_off[2][2];   // [half][piece]
    unsigned a_taps[2][2];               // CONV3: bit (3 r + q) = tap (r, q) of this row's pixel lies inside the image (0 for rows past M)
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = __builtin_amdgcn_rcpf((float)HoWo), inv_wo = __builtin_amdgcn_rcpf((float)p.Wo);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = h * 128 + (8 * i + wave) * 8 + lrow;
            const int m = m0 + r;
            if constexpr (CONV3) {
                const bool ok = m < p.M;
                const int mm = ok ? m : 0;
                int n = (int)((float)mm * inv_howo);   // small quotients: a float estimate is within +-1, fixed up exactly
                int rem = mm - n * HoWo;
                if (rem < 0) { --n; rem += HoWo; }
                if (rem >= HoWo) { ++n; rem -= HoWo; }
                int ho = (int)((float)rem * inv_wo);
                int wo = rem - ho * p.Wo;
                if (wo < 0) { --ho; wo += p.Wo; }
                if (wo >= p.Wo) { ++ho; wo -= p.Wo; }
                // byte offset of (n, ho - 1, wo - 1, channel chunk): wraps for border pixels, whose taps are masked
                a_off[h][i] = (unsigned)(((n * p.H + ho - 1) * p.W + wo - 1) * lda + lchunk * 8) * 2u;
                unsigned rows = 0, cols = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    rows |= ((unsigned)(ho - 1 + t) < (unsigned)p.H ? 1u : 0u) << t;
                    cols |= ((unsigned)(wo - 1 + t) < (unsigned)p.W ? 1u : 0u) << t;
                }
                unsigned taps = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) taps |= ((rows >> t) & 1u) ? (cols << (3 * t)) : 0u;
                a_taps[h][i] = ok ? taps : 0u;
            } else {
                a_off[h][i] = m < p.M ? (unsigned)(((size_t)m * lda + lchunk * 8) * 2) : kOOB;
                a_taps[h][i] = 0;
            }
            w_off[h][i] = (n0 + r) < p.Cout_pad ? (unsigned)(((size_t)(n0 + r) * ldw + lchunk * 8) * 2) : kOOB;
        }
    const int spt = CONV3 ? p.CinK / BK : 1;                       // K-tiles per tap
    const unsigned spt_m = CONV3 ? (65536u + spt - 1) / spt : 0u;   // kt / spt == (kt * spt_m) >> 16 for kt < 65536 / spt (conv_gemm256_possible)
    // half-tile id: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi; K-tiles beyond nk are range-checked away (zero fill)
    auto stage = [&](int buf, int hid, int kt) {
        const bool live = kt < nk;
        const unsigned koff = (unsigned)kt * (BK * 2);
        if (CONV3 && hid < 2) {   // an A half-tile of the 3x3: K-tile kt = (tap, 64-channel slice)
            const int tap = (int)(((unsigned)kt * spt_m) >> 16);
            const int cs = kt - tap * spt;
            const int r = (tap * 11) >> 5, q = tap - 3 * r;   // tap / 3 for tap < 12
            const unsigned delta = (unsigned)((r * p.W + q) * lda * 2 + cs * (BK * 2));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                char* dst = smem + buf * BUF_B + hid * HALF_B + (8 * i + wave) * 1024;
                const bool ok = live && ((a_taps[hid & 1][i] >> tap) & 1u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)dst, 16, ok ? a_off[hid & 1][i] + delta : kOOB, 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char* dst = smem + buf * BUF_B + hid * HALF_B + (8 * i + wave) * 1024;
            const unsigned base = hid < 2 ? a_off[hid & 1][i] : w_off[hid & 1][i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(hid < 2 ? rs_a : rs_w, (lds_ptr_t)dst, 16, live ? base : kOOB, koff, 0, 0);
        }
    };

    // fragment read offsets (16 x 16 x 32: two k-slices per K-tile, logical chunks lane >> 4 and (lane >> 4) + 4)
    const int frow = lane & 15;
    int f_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) f_off[h] = frow * ROW_B + ((((lane >> 4) + h * 4) ^ ((frow >> 1) & 7)) * 16);
    const int a_base = wm * HALF_B;                                               // this wave's 128 rows = one A half-tile
    const int b_base = 2 * HALF_B + (wn >> 1) * HALF_B + (wn & 1) * 64 * ROW_B;   // its 64 columns = half of a B half-tile

    floatx4 acc[8][4];
    half8 ra[2][4], rb[2][2][2];
    auto read_a = [&](int buf, int s) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[h][i] = *reinterpret_cast<const half8*>(smem + buf * BUF_B + a_base + (s * 64 + i * 16) * ROW_B + f_off[h]);
    };
    auto read_b = [&](int buf, int s) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) rb[s][h][j] = *reinterpret_cast<const half8*>(smem + buf * BUF_B + b_base + (s * 32 + j * 16) * ROW_B + f_off[h]);
    };
    auto mma = [&](int sa, int sb) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[sa * 4 + i][sb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rb[sb][h][j], ra[h][i], acc[sa * 4 + i][sb * 2 + j], 0, 0, 0);
    };

#define G256_LOAD_END(WAIT)                                                       \
    if (WAIT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                    \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0), told to the compiler */    \
    __builtin_amdgcn_sched_barrier(0);                                            \
    __builtin_amdgcn_s_barrier();                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define G256_COMPUTE_END()                                                        \
    __builtin_amdgcn_sched_barrier(0);                                            \
    __builtin_amdgcn_s_barrier();                                                 \
    __builtin_amdgcn_sched_barrier(0);

    // prologue: B(E0), A(E0), B(O1) - the six half-tiles the first phases read or the schedule does not stage itself
    stage(0, 2, 0); stage(0, 3, 0); stage(0, 0, 0); stage(0, 1, 0); stage(1, 2, 1); stage(1, 3, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (wm == 1) __builtin_amdgcn_s_setprio(1);
    if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one segment behind group 0

#define G256_PHASE(READS, BUF, HID, KT, WAIT, SA, SB)  \
    {                                                  \
        READS;                                         \
        stage(BUF, HID, KT);                           \
        G256_LOAD_END(WAIT);                           \
        mma(SA, SB);                                   \
        G256_COMPUTE_END();                            \
    }
    for (int kt = 0; kt < nk; kt += 2) {
        G256_PHASE(read_b(0, 0); read_a(0, 0), 1, 0, kt + 1, false, 0, 0);
        G256_PHASE(read_b(0, 1), 1, 1, kt + 1, false, 0, 1);
        G256_PHASE(read_a(0, 1), 0, 2, kt + 2, false, 1, 1);
        G256_PHASE((void)0, 0, 3, kt + 2, true, 1, 0);
        G256_PHASE(read_b(1, 0); read_a(1, 0), 0, 0, kt + 2, false, 0, 0);
        G256_PHASE(read_b(1, 1), 0, 1, kt + 2, false, 0, 1);
        G256_PHASE(read_a(1, 1), 1, 2, kt + 3, false, 1, 1);
        G256_PHASE((void)0, 1, 3, kt + 3, true, 1, 0);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-out pieces (range-checked away) still write zeros into LDS
#undef G256_PHASE
#undef G256_LOAD_END
#undef G256_COMPUTE_END

    // ---- epilogue.  Accumulator side (lane = 4 consecutive channels of pixel lane & 15): + bias in fp32, act1, ONE rounding to fp16, into a
    // wave-private LDS chunk (64 rows x 64 channels fp16, 144-byte rows); row-major side (lane = 8 consecutive channels of one pixel): the
    // residual added to the ROUNDED value, act2, one 16-byte store - 8 lanes write one 128-byte line.
    __builtin_amdgcn_s_barrier();   // every wave has retired its fragment reads and its DMA writes: the stage buffers are free
    constexpr int EPS = 144;
    char* mine = smem + wave * 64 * EPS;
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    float bias[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
        const float4 b4 = (p.bias && n < p.Cout_pad) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        bias[j][0] = b4.x; bias[j][1] = b4.y; bias[j][2] = b4.z; bias[j][3] = b4.w;
    }
    const bool second = res || p.act2 != ACT_NONE;
    half8 rvs[2][8];
    auto fetch_shortcut = [&](int c) {
        if (!res) return;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = m0 + wm * 128 + c * 64 + it * 8 + (lane >> 3);
            const int n = n0 + wn * 64 + (lane & 7) * 8;
            const int mc = m < p.M ? m : p.M - 1, nc = n < p.Cout ? n : p.Cout - 8;
            rvs[c][it] = *reinterpret_cast<const half8*>(res + (size_t)mc * p.ld_res + nc);
        }
    };
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        // the shortcut values of this 64-row chunk, ALL EIGHT fetched before anything waits for one (unconditional loads from clamped
        // rows): inside the store loop below each load sat in front of its use - 16 dependent HBM round trips per tile, a third of the
        // 512 -> 2048 GEMM's time (round 4: 780 us with them in the loop against 498-535 us for the harness without a shortcut)
        // Round 5: the SECOND chunk's eight are fetched in front of the first chunk's stores too (fetch(1) below) - issued after them, the wait for the shortcut
        // values was a wait for the stores' acknowledgements as well (tools/isa_store_waits.py: 4 of a wave's 16 stores stood behind a full vmcnt(0)).
        if (c == 0) fetch_shortcut(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const floatx4 v = acc[c * 4 + i][j];
                half4 h;
                if (p.act1 == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = v[e] + bias[j][e];
                        h[e] = round_to_half(x > 0.f ? x : 0.f);
                    }
                } else if (p.act1 == ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = round_to_half(v[e] + bias[j][e]);
                } else if (p.act1 == ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = v[e] + bias[j][e];
                        h[e] = round_to_half(x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = round_to_half(g256_act_slow(v[e] + bias[j][e], p.act1, p.alpha1));
                }
                *reinterpret_cast<half4*>(mine + (i * 16 + (lane & 15)) * EPS + (j * 16 + 4 * (lane >> 4)) * 2) = h;
            }
        if (c == 0) fetch_shortcut(1);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int m = m0 + wm * 128 + c * 64 + row;
            const int n = n0 + wn * 64 + (lane & 7) * 8;
            half8 v = *reinterpret_cast<const half8*>(mine + row * EPS + (lane & 7) * 16);
            if (m >= p.M || n >= p.Cout) continue;
            if (second) {
                half8 rv = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (res) rv = rvs[c][it];
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
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_to_half(g256_act((float)v[e] + (float)rv[e], p.act2, p.alpha2));
                }
            }
            *reinterpret_cast<half8*>(out + (size_t)m * p.ld_out + n) = v;
        }
    }
}

}  // namespace

// fp16 plain GEMM (1x1, stride 1, unpadded, Cin == CinK), K a whole number of 64-wide tiles and at least 4 of them, Cout a whole number of 256-wide
// tiles (the packed weights have exactly Cout_pad rows), 16-byte rows everywhere, the addressed slices below 2 GB.  This is the LAUNCH-time predicate: it holds
// at every batch up to the one the engine was built for (it used to contain the profitability rule below, which depends on the runtime batch - a
// layer the tuner had put on this tile at max_batch was then refused at a smaller batch and the whole enqueue failed).
bool conv_gemm256_possible(const ConvArgs& a) {
    if (a.in_i8 || a.out_i8 || a.res_i8 || a.up_C || a.scalar_out || a.groups != 1) return false;
    const bool plain = a.kh == 1 && a.kw == 1 && a.stride_h == 1 && a.stride_w == 1 && a.pad_h == 0 && a.pad_w == 0 && a.Cin == a.CinK && a.Kpad == a.K;
    // ... or a 3x3 stride-1 pad-1 convolution over whole 64-channel K-tiles (res5's 512 -> 512 on 4000 RoIs, rcnn/backbone.hpp:100-229)
    const bool conv3 = a.kh == 3 && a.kw == 3 && a.stride_h == 1 && a.stride_w == 1 && a.pad_h == 1 && a.pad_w == 1 && a.dil_h == 1 && a.dil_w == 1 &&
                       a.Cin == a.CinK && a.CinK % 64 == 0 && a.Kpad == 9 * a.CinK && a.Ho == a.H && a.Wo == a.W && 9 * (a.CinK / 64) * (a.CinK / 64) < 60000 &&
                       (double)a.N * a.H * a.W < 8.0e6;   // (pixel index split with float reciprocals; kt / spt by a 16-bit reciprocal)
    if (!plain && !conv3) return false;
    if (a.Kpad % 64 || a.Kpad < 256 || a.Cout_pad % 256 || a.Cout % 8 || a.ld_in % 8 || a.ld_out % 8 || (a.residual && a.ld_res % 8)) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    if ((double)M * a.ld_in * 2.0 >= 2.0e9 || (double)a.Cout_pad * a.Kpad * 2.0 >= 2.0e9) return false;
    return true;
}

// ... and the rule that makes it a CANDIDATE of the tactic timing (conv_tactics(), at the batch the engine is built for): from 96 tiles on - one tile per CU on a
// third of the chip; whether it beats the 128-row tiles there is the timing's call
bool conv_gemm256_worthwhile(const ConvArgs& a) {
    const long M = (long)a.N * a.Ho * a.Wo;
    return conv_gemm256_possible(a) && ((M + 255) / 256) * (a.Cout_pad / 256) >= 96;
}

int32_t conv_gemm256_f16(const ConvArgs& a0, hipStream_t s) {
    if (!conv_gemm256_possible(a0)) return TRTX_ERR_UNSUPPORTED;
    ConvArgs a = a0;
    a.M = a.N * a.Ho * a.Wo;
    const unsigned a_bytes = (unsigned)((((size_t)a.M - 1) * a.ld_in + a.Cin) * 2);
    const unsigned w_bytes = (unsigned)((size_t)a.Cout_pad * a.Kpad * 2);
    const int tiles_n = a.Cout_pad / 256, total = ((a.M + 255) / 256) * tiles_n, chunk = (total + 7) / 8;
    if (a.kh == 3) TRTX_LAUNCH(conv_gemm256_f16_kernel<true>, dim3(chunk * 8), dim3(512), 0, s, a, a_bytes, w_bytes, tiles_n, total, chunk);
    else TRTX_LAUNCH(conv_gemm256_f16_kernel<false>, dim3(chunk * 8), dim3(512), 0, s, a, a_bytes, w_bytes, tiles_n, total, chunk);
    return check_launch("conv_gemm256_f16");
}

}  // namespace trtx
