// Weight-stationary persistent convolution for gfx950 (MI355X): the kernel behind the small-channel 3x3 / 1x1 layers that
// make up YOLOv8 (yolov8/src/block.cpp:79-155 convBnSiLU / bottleneck / C2F, model.cpp:188-251 detect heads), where the
// implicit-GEMM kernel of conv_igemm.hip is bound by instruction issue and by the L1 -> LDS path (every input pixel crosses it
// once per filter tap, 9x for a 3x3).
//
// Design (DESIGN.md "conv_ws"):
//   * one persistent workgroup per CU (4 waves, one per SIMD, up to 512 VGPRs each) loops over output tiles of 64 or 128 pixels;
//   * the WEIGHTS LIVE IN REGISTERS for the whole kernel: wave (wp, wc) owns NFW 16-channel column fragments x all K as MFMA B
//     operands (<= 288 VGPRs), loaded once from L2;
//   * the INPUT PATCH (tile + halo) of a tile is brought HBM/L2 -> LDS exactly once by buffer_load ... lds (zero fill of the
//     padding ring and of ragged channels through the buffer range check), double buffered: the patch of tile t+1 streams in
//     while tile t is computed, ONE workgroup barrier per tile;
//   * the k-loop is ds_read_b128 (A fragment at a precomputed per-lane address + immediate) -> NFW x v_mfma_f32_16x16x32_f16,
//     fully unrolled: no address arithmetic, no global loads, no barriers, no waits inside;
//   * LDS layout: one plane per 32-channel slice, 64 B per pixel, the four 16-B chunks of a pixel XOR-swizzled by
//     (pixel >> 1) & 3 — conflict free for any run of 16 consecutive pixels and for tile rows whose pitch gap is a multiple of
//     8 pixels (brute-forced over the ds_read_b128 lane groups of MI355X_MICROARCH.md);
//   * epilogue: bias(BN) -> act1 -> (+residual) -> act2 in registers, 8-byte stores into the (possibly strided) NHWC slice.
// Geometry: 3x3 stride 1 pad 1 uses TH x TW tiles of one image; 1x1 stride 1 treats the tensor as one row of N*H*W pixels.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>

#include "../common.h"
#include "../options.h"
#include "kernels.h"
#include "launch.h"

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr unsigned kOOB = 0x80000000u;  // beyond any num_records: the buffer load returns 0 and touches no memory

struct WsGeom {
    int TH, TW;          // output tile (TH * TW <= 16 * NF pixels)
    int PH, PW, PWp;     // input patch rows / columns / LDS row pitch in pixels ((PWp - TW) % 8 == 0 when TW is not 16)
    int tiles_x, tiles_y, total_tiles;
    int plane_bytes;     // LDS bytes of one 32-channel plane of the patch (multiple of 1024)
    int n_dma;           // buffer_load...lds instructions per plane (16 pixels each)
    int xcd_chunk;       // tiles per XCD in the XCD-aware tile order
    int grid;            // workgroups launched
    float inv_tw, inv_pwp, inv_tpi;  // reciprocals for the exact small-integer divisions
    int dbg;             // timing experiments (TRTX_WS_DBG): 1 no patch DMA, 2 no MFMA loop, 4 no stores, 8 no weight loads
};

__device__ __forceinline__ float act_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// the rare activation kinds, out of line (one copy in the kernel instead of one per call site: code size is a cost here)
__device__ __attribute__((noinline)) float ws_act_rare(float v, int act, float alpha) {
    if (act == ACT_LEAKY) return v > 0.f ? v : v * alpha;
    if (act == ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (act == ACT_TANH) return tanhf(v);
    if (act == ACT_MISH) return mish_ref(v);
    return v;
}
__device__ __forceinline__ float ws_act_any(float v, int act, float alpha) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU) return act_silu(v);
    if (act == ACT_NONE) return v;
    return ws_act_rare(v, act, alpha);
}
// exact x / d for 0 <= x < 2^22 with a float reciprocal estimate fixed up by one step
__device__ __forceinline__ int div_small(int x, int d, float inv) {
    int q = (int)((float)x * inv);
    int r = x - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; }
    return q;
}

// TAPS: 1 (1x1) or 9 (3x3); KC: 32-channel slices of Cin; NFW: 16-channel output fragments per wave; WC: waves that split the
// output channels (1, 2, 4; the other 4 / WC waves split the tile's pixel fragments); NF: 16-pixel fragments per tile (4 or 8).
// ROWS: the tile is NF rows of 16 pixels and the LDS row pitch is a multiple of 8 pixels, so fragment f's addresses are fragment
// 0's plus f * pitch (the swizzle term does not change): TAPS address registers instead of AFW * TAPS.
// Everything a wave keeps live (weights KS*NFW*4, accumulators AFW*NFW*4, addresses) is sized to stay inside the 256
// architectural VGPRs: with more, the compiler parks operands in AGPRs and the k-loop fills up with v_accvgpr moves.
// registers a wave keeps live: stationary weights + accumulators + A addresses + two sets of A fragments + bias + ~28 of bookkeeping
constexpr int ws_regs(int taps, int kc, int nfw, int wc, int nf, bool rows) {
    const int afw = nf / (4 / wc);
    const int fg = afw < 4 ? afw : 4;
    return taps * kc * nfw * 4 + afw * nfw * 4 + (rows ? 1 : afw) * (taps + 2) + 2 * fg * 4 + nfw * 4 + 28;
}
// waves per SIMD the register budget allows (512 / waves registers each): more resident workgroups = more bytes in flight
constexpr int ws_min_waves(int regs) { return regs <= 120 ? 4 : (regs <= 244 ? 2 : 1); }

template <int TAPS, int KC, int NFW, int WC, int NF, bool ROWS>
__global__ __launch_bounds__(256, ws_min_waves(ws_regs(TAPS, KC, NFW, WC, NF, ROWS))) void conv_ws_f16_kernel(const ConvArgs p, const WsGeom g,
                                                                                                           unsigned in_bytes) {
    constexpr int WP = 4 / WC;
    constexpr int AFW = NF / WP;   // pixel fragments per wave
    constexpr int KW = TAPS == 9 ? 3 : 1;
    constexpr int KS = TAPS * KC;  // 32-wide k-steps
    static_assert(AFW >= 1, "tile too small for this wave split");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % WP, wc = wave / WP;
    const int buf_bytes = KC * g.plane_bytes;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, in_bytes, 0x00020000);

    // XCD-aware persistent tile order: workgroup id -> (xcd = id % 8, slot = id / 8); each XCD walks a contiguous range
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = g.grid >> 3;
    const int tile_end = min((xcd + 1) * g.xcd_chunk, g.total_tiles);
    int tile = xcd * g.xcd_chunk + slot;
    const int tpi = g.tiles_x * g.tiles_y;

    auto issue_patch = [&](int t, int b) {
        if (g.dbg & 1) return;
        // tile -> (image, tile row, tile column); wave-uniform
        const int n = div_small(t, tpi, g.inv_tpi);
        const int r = t - n * tpi;
        const int ty = r / g.tiles_x, tx = r - ty * g.tiles_x;
        const int hi0 = ty * g.TH - p.pad_h, wi0 = tx * g.TW - p.pad_w;
        const int img = n * p.H;
        char* base = smem + b * buf_bytes;
        for (int jj = wave; jj < g.n_dma; jj += 4) {
            const int pp = jj * 16 + (lane >> 2);
            const int clog = (lane & 3) ^ ((pp >> 1) & 3);  // this lane fills physical chunk (lane & 3) of pixel pp
            const int py = div_small(pp, g.PWp, g.inv_pwp);
            const int px = pp - py * g.PWp;
            const int hi = hi0 + py, wi = wi0 + px;
            const bool ok = py < g.PH && px < g.PW && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned pix = (unsigned)(((img + hi) * p.W + wi) * p.ld_in + clog * 8) * 2u;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const bool okc = ok && (kc * 32 + clog * 8 < p.Cin);
                const unsigned voff = okc ? pix + kc * 64u : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(base + kc * g.plane_bytes + jj * 1024), 16, voff, 0, 0, 0);
            }
        }
    };

    if (tile < tile_end) issue_patch(tile, 0);

    // ---- weights -> registers (once).  Packed layout [Cout_pad][Kpad], k = tap * CinK + c, CinK = 32 * KC.
    half8 breg[KS][NFW];
    float4 bias4[NFW];
    {
        const _Float16* w = static_cast<const _Float16*>(p.wgt);
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            const int row = (wc * NFW + j) * 16 + (lane & 15);
            const bool live = row < p.Cout_pad;
            const _Float16* wr = w + (size_t)(live ? row : 0) * p.Kpad + (lane >> 4) * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                half8 v = (g.dbg & 8) ? half8{1, 1, 1, 1, 1, 1, 1, 1} : *reinterpret_cast<const half8*>(wr + ks * 32);
                if (!live) v = half8{0, 0, 0, 0, 0, 0, 0, 0};
                breg[ks][j] = v;
            }
            const int co = (wc * NFW + j) * 16 + (lane >> 4) * 4;
            bias4[j] = (p.bias && co < p.Cout_pad) ? *reinterpret_cast<const float4*>(p.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // ---- per-lane A-fragment addresses inside a patch buffer (tile independent).  Fragment f of this wave covers tile pixels
    // t = (wp * AFW + f) * 16 + (lane & 15); lane reads 32-channel slice chunk (lane >> 4) of pixel (oy + r, ox + q).
    constexpr int AF_ADDR = ROWS ? 1 : AFW;
    unsigned a_addr[AF_ADDR][TAPS];
#pragma unroll
    for (int f = 0; f < AF_ADDR; ++f) {
        const int t = (wp * AFW + f) * 16 + (lane & 15);
        const bool live = t < g.TH * g.TW;
        const int tt = live ? t : 0;
        const int oy = ROWS ? wp * AFW : div_small(tt, g.TW, g.inv_tw);
        const int ox = ROWS ? (lane & 15) : tt - oy * g.TW;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int r = tap / KW, q = tap % KW;
            const int pp = (oy + r) * g.PWp + ox + q;
            a_addr[f][tap] = (unsigned)(pp * 64 + (((lane >> 4) ^ ((pp >> 1) & 3)) << 4));
        }
    }
    const int row_pitch_bytes = g.PWp * 64;  // ROWS: fragment f = tile row wp * AFW + f

    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    const bool second = res || p.act2 != ACT_NONE;

    int b = 0;
    for (; tile < tile_end; tile += per_xcd, b ^= 1) {
        // my DMA for this tile has landed (and my stores of the previous tile have been issued long ago) ...
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (lgkmcnt: my reads of the other buffer have COMPLETED - conv_igemm.hip, k-step comment)
        // ... so has everybody's, and every wave is done reading the other buffer
        __builtin_amdgcn_s_barrier();
        const int next = tile + per_xcd;
        if (next < tile_end) issue_patch(next, b ^ 1);

        const char* buf = smem + b * buf_bytes;
        floatx4 acc[AFW][NFW];
#pragma unroll
        for (int f = 0; f < AFW; ++f)
#pragma unroll
            for (int j = 0; j < NFW; ++j) acc[f][j] = floatx4{0.f, 0.f, 0.f, 0.f};

        // k-loop.  One wave per SIMD (or two) has nobody else to hide latencies behind, so the loop itself does: fragments are
        // processed FG at a time (FG * NFW independent accumulator chains keep the MFMA pipe full) and the FG A fragments of
        // k-step ks+1 are in flight (ds_read_b128) during the FG * NFW MFMAs of k-step ks.
        constexpr int FG = AFW < 4 ? AFW : 4;
        auto a_load = [&](int f, int ks) -> half8 {
            const int tap = ks / KC, kc = ks % KC;
            if (ROWS) return *reinterpret_cast<const half8*>(buf + a_addr[0][tap] + f * row_pitch_bytes + kc * g.plane_bytes);
            return *reinterpret_cast<const half8*>(buf + a_addr[f][tap] + kc * g.plane_bytes);
        };
#pragma unroll
        for (int f0 = 0; f0 < AFW && !(g.dbg & 2); f0 += FG) {
            half8 a_cur[FG], a_nxt[FG];
#pragma unroll
            for (int i = 0; i < FG; ++i) a_cur[i] = a_load(f0 + i, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int i = 0; i < FG; ++i) a_nxt[i] = a_load(f0 + i, ks + 1);
                }
#pragma unroll
                for (int j = 0; j < NFW; ++j)
#pragma unroll
                    for (int i = 0; i < FG; ++i)
                        acc[f0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(breg[ks][j], a_cur[i], acc[f0 + i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < FG; ++i) a_cur[i] = a_nxt[i];
            }
        }

        // ---- epilogue: lane holds channels co .. co+3 of pixel t of fragment f
        const int n = div_small(tile, tpi, g.inv_tpi);
        const int rr = tile - n * tpi;
        const int ty = rr / g.tiles_x, tx = rr - ty * g.tiles_x;
        // act1 in registers, then through a wave-private LDS tile so that global traffic is row-major 16-byte chunks: residual
        // reads and output stores cover whole pixels (64..256 B runs) instead of 8-byte slivers of 16 different lines
        // (measured on the streaming 1x1 layers: the sliver stores alone cost 6-9 us of a 16-21 us kernel).
        constexpr int RS = NFW * 32 + 16;        // staging row stride: 16 consecutive rows start in distinct bank groups
        constexpr int CPR = NFW * 2;             // 16-byte chunks per staged row
        constexpr int ITEMS = 16 * CPR;
        char* stg = smem + 2 * buf_bytes + wave * (16 * RS);
        // The activation kind is a wave-uniform RUN-TIME branch around each small group of elements, and the row-major pass is a
        // rolled loop: the finishing code exists once per accumulator fragment instead of once per (fragment, act1, act2)
        // combination - a sixth of the instructions, and the kernel stays inside the instruction cache next to its neighbours
        // (see the epilogue of conv_igemm_f16_kernel).
#pragma unroll
        for (int f = 0; f < AFW; ++f) {
#pragma unroll
            for (int j = 0; j < NFW; ++j) {
                const float b4[4] = {bias4[j].x, bias4[j].y, bias4[j].z, bias4[j].w};
                half4 o;
                if (p.act1 == ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = round_to_half(act_silu(acc[f][j][e] + b4[e]));
                } else if (p.act1 == ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = round_to_half(acc[f][j][e] + b4[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = round_to_half(ws_act_any(acc[f][j][e] + b4[e], p.act1, p.alpha1));
                }
                *reinterpret_cast<half4*>(stg + (lane & 15) * RS + j * 32 + (lane >> 4) * 8) = o;
            }
#pragma nounroll
            for (int q = lane; q < ITEMS; q += 64) {
                const int row = q / CPR, cc = q - row * CPR;
                // the pixel this lane now stores: row `row` of fragment f
                int oy, ox;
                if (ROWS) {
                    oy = wp * AFW + f;
                    ox = row;
                } else {
                    const int t = (wp * AFW + f) * 16 + row;
                    oy = div_small(t, g.TW, g.inv_tw);
                    ox = t - oy * g.TW;
                }
                const int ho = ty * g.TH + oy, wo = tx * g.TW + ox;
                const int co = wc * NFW * 16 + cc * 8;
                if (oy >= g.TH || ho >= p.Ho || wo >= p.Wo || co >= p.Cout || (g.dbg & 4)) continue;
                const size_t m = (size_t)(n * p.Ho + ho) * p.Wo + wo;
                half8 v = *reinterpret_cast<const half8*>(stg + row * RS + cc * 16);
                if (second) {
                    half8 rv = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (res) rv = *reinterpret_cast<const half8*>(res + m * p.ld_res + co);
                    if (p.act2 == ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = round_to_half((float)v[e] + (float)rv[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = round_to_half(ws_act_any((float)v[e] + (float)rv[e], p.act2, p.alpha2));
                    }
                }
                *reinterpret_cast<half8*>(out + m * p.ld_out + co) = v;
            }
        }
    }
}

struct WsConfig {
    int taps, kc, nfw, wc, nf, rows;
};

constexpr int kLdsBudget = 160 * 1024;
constexpr int kRegBudget = 256;  // ws_regs() of an entry: everything must fit the 256 architectural VGPRs

// (TAPS, KC, NFW, WC, NF, ROWS) combinations that are instantiated.  -DTRTX_WS_ONLY="X(9, 2, 2, 2, 8, 1)" builds a subset
// (kernel development: each instantiation is a fully unrolled k-loop and takes a few seconds to compile).
#ifdef TRTX_WS_ONLY
#define TRTX_WS_CASES(X) TRTX_WS_ONLY
#else
#define TRTX_WS_CASES(X)                                                          \
    X(9, 1, 2, 1, 8, 1) /* 3x3 32 -> 32 on 16-pixel row tiles (C2f bottlenecks) */ \
    X(1, 1, 2, 1, 8, 0) /* 1x1 32 -> 32 streaming                              */
// Why only Cin = 32 (profiles/r02_trace_gaps_*.txt, r02_ws_per_op.txt): every workgroup starts by pulling the layer's whole
// weight set into registers (18 KB here, 73 KB for 64 -> 64 3x3).  Timed one op at a time the wider configurations also beat the
// implicit-GEMM kernel (64 -> 64 3x3: 94 vs 116 us at 320^2 b8), but inside the engine's back-to-back stream the first launch of
// each of them costs 7-12 us more than it saves (conv_ws<9,2,2,2,8> 46.7 us against 34.7 for conv_igemm, then 31.2 on the
// second launch), so YOLOv8n / ResNet-50 / RetinaFace / R-CNN steps were 0.6-2.7 % slower with them enabled.  The template
// still supports those shapes (KC <= 12, WC 1/2/4, 64-pixel tiles): build with -DTRTX_WS_ONLY="X(...)" to experiment.
#endif

struct WsEntry {
    int taps, kc, nfw, wc, nf, rows;
};
const WsEntry kWsTable[] = {
#define X(T, K, N, W, F, R) {T, K, N, W, F, R},
    TRTX_WS_CASES(X)
#undef X
};

// tile geometry of a table entry for this layer; false when it does not apply (LDS, shape)
bool ws_geometry(const ConvArgs& a, const WsEntry& e, WsGeom* o) {
    const int px = e.nf * 16;
    const long M = (long)a.N * a.Ho * a.Wo;
    if (e.taps == 1) {
        if (e.rows) return false;
        o->TH = 1; o->TW = px; o->PH = 1; o->PW = px; o->PWp = px;
        o->tiles_y = 1; o->tiles_x = (int)((M + px - 1) / px);
        o->total_tiles = o->tiles_x;
    } else {
        int tw, th, pwp;
        if (e.rows) {  // NF rows of 16 pixels, pitch a multiple of 8 pixels
            if (a.Wo < 16) return false;
            tw = 16; th = e.nf; pwp = 24;
        } else {       // whole (short) rows: TW = W, fragments run across rows, row-break gap a multiple of 8 pixels
            if (a.Wo > px || a.Wo > 64) return false;
            tw = a.Wo; th = std::max(1, px / tw); pwp = tw + 8;
        }
        th = std::min(th, a.Ho);
        o->TH = th; o->TW = tw; o->PH = th + 2; o->PW = tw + 2; o->PWp = pwp;
        o->tiles_y = (a.Ho + th - 1) / th; o->tiles_x = (a.Wo + tw - 1) / tw;
        o->total_tiles = a.N * o->tiles_y * o->tiles_x;
    }
    o->n_dma = (o->PH * o->PWp + 15) / 16;
    o->plane_bytes = o->n_dma * 1024;
    if (2L * e.kc * o->plane_bytes + 4 * 16 * (e.nfw * 32 + 16) > kLdsBudget) return false;  // two patch buffers + epilogue staging
    const int afw = e.nf / (4 / e.wc);
    if (afw < 1) return false;
    return ws_regs(e.taps, e.kc, e.nfw, e.wc, e.nf, e.rows != 0) <= kRegBudget;
}

bool pick_config(const ConvArgs& a, WsConfig* c, WsGeom* g) {
    const int taps = a.kh * a.kw;
    if (!((a.kh == 1 && a.kw == 1 && a.pad_h == 0 && a.pad_w == 0) || (a.kh == 3 && a.kw == 3 && a.pad_h == 1 && a.pad_w == 1))) return false;
    if (a.stride_h != 1 || a.stride_w != 1 || a.dil_h != 1 || a.dil_w != 1 || a.groups != 1) return false;
    if (a.bk != 32 || a.CinK % 32 != 0 || a.CinK < a.Cin || a.Kpad != taps * a.CinK) return false;
    if (a.Cin % 8 || a.ld_in % 8 || a.Cout % 8 || a.ld_out % 8 || (a.residual && a.ld_res % 8) || a.scalar_out) return false;
    if (a.Ho != a.H || a.Wo != a.W) return false;
    const int kc = a.CinK / 32;
    const int nfrag = (a.Cout + 15) / 16;
    // best entry: fewest wasted MFMA columns, then useful pixels per tile slot, then enough tiles to fill 256 CUs
    double best = -1.0;
    for (const WsEntry& e : kWsTable) {
        if (e.taps != taps || e.kc != kc || e.nfw * e.wc < nfrag) continue;
        WsGeom t{};
        if (!ws_geometry(a, e, &t)) continue;
        const double col_eff = (double)nfrag / (e.nfw * e.wc);
        const double px_eff = (double)a.N * a.Ho * a.Wo / ((double)t.total_tiles * e.nf * 16);
        const double rounds = (double)t.total_tiles / 256.0;
        const double fill = rounds >= 1.0 ? rounds / std::ceil(rounds) : rounds;  // last-round / under-subscription loss
        const double score = col_eff * px_eff * fill * (e.wc == 1 ? 1.0 : 0.97);
        if (score > best) {
            best = score;
            *c = WsConfig{e.taps, e.kc, e.nfw, e.wc, e.nf, e.rows};
            *g = t;
        }
    }
    if (best < 0) return false;
    // The stationary weights are loaded once per WORKGROUP: that only pays when a workgroup then walks several tiles.  Measured
    // (tools/gpu_ws_probe.sh, batch 32): 80x80 and 160x160 maps (1600+ tiles) win 15-30 % over the implicit-GEMM kernel,
    // 40x40 / 20x20 maps (<= 450 tiles, about one per workgroup) lose 20-40 %.
    constexpr int min_tiles = 1024;
    if (g->total_tiles < min_tiles) return false;
    const int tpi = g->tiles_x * g->tiles_y;
    g->inv_tw = 1.0f / (float)g->TW;
    g->inv_pwp = 1.0f / (float)g->PWp;
    g->inv_tpi = 1.0f / (float)tpi;
    g->grid = (std::min(g->total_tiles, 256) + 7) / 8 * 8;
    g->xcd_chunk = (g->total_tiles + 7) / 8;
    return true;
}

template <int TAPS, int KC, int NFW, int WC, int NF, bool ROWS>
int32_t launch_ws(const ConvArgs& a, const WsGeom& g, unsigned in_bytes, hipStream_t s, int* occ_cache) {
    auto kern = conv_ws_f16_kernel<TAPS, KC, NFW, WC, NF, ROWS>;
    const size_t lds = 2 * (size_t)KC * g.plane_bytes + 4 * 16 * (NFW * 32 + 16);
    static std::atomic<bool> attr_done[64];  // per instantiation and per device (one process may drive several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TRTX_ERR_HIP;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget) != hipSuccess) {
            (void)hipGetLastError();
            return TRTX_ERR_HIP;
        }
        attr_done[dev].store(true, std::memory_order_release);
    }
    // persistent grid: as many workgroups as stay resident (registers and this launch's LDS), on 256 CUs
    int occ = *occ_cache;
    if (occ <= 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess || occ < 1) {
            (void)hipGetLastError();
            occ = 1;
        }
        constexpr int occ_cap = 4;
        occ = std::min(occ, occ_cap);
        *occ_cache = occ;
    }
    WsGeom gg = g;
    gg.dbg = 0;
    gg.grid = (std::min(g.total_tiles, 256 * occ) + 7) / 8 * 8;
    TRTX_LAUNCH(kern, dim3(gg.grid), dim3(256), lds, s, a, gg, in_bytes);
    return TRTX_OK;
}

int32_t dispatch_ws(const ConvArgs& a, const WsConfig& c, const WsGeom& g, unsigned in_bytes, hipStream_t s, int* occ_cache) {
#define X(T, K, N, W, F, R) \
    if (c.taps == T && c.kc == K && c.nfw == N && c.wc == W && c.nf == F && c.rows == R) return launch_ws<T, K, N, W, F, (R != 0)>(a, g, in_bytes, s, occ_cache);
    TRTX_WS_CASES(X)
#undef X
    return TRTX_ERR_UNSUPPORTED;
}

// Launch plans are a pure function of the layer shape: computed once, looked up per launch (the executor enqueues ~60 convs per
// step from one host thread; nothing shape-dependent is recomputed on that path).
struct WsPlan {
    bool ok = false;
    WsConfig c{};
    WsGeom g{};
    int occ = 0;
};
struct WsKey {
    int v[16];
    bool operator<(const WsKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};

WsPlan* ws_plan(const ConvArgs& a) {
    static std::mutex mu;
    static std::map<WsKey, WsPlan> cache;
    const WsKey k{{a.N, a.H, a.W, a.Cin, a.ld_in, a.Cout, a.Cout_pad, a.ld_out, a.residual ? a.ld_res : -1, a.kh * 16 + a.kw,
                   a.stride_h * 16 + a.stride_w, a.pad_h * 16 + a.pad_w, a.dil_h * 16 + a.dil_w + 256 * a.groups, a.CinK, a.Kpad,
                   a.bk * 2 + a.scalar_out}};
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(k);
    if (it == cache.end()) {
        WsPlan p;
        const double in_b = (double)a.N * a.H * a.W * a.ld_in * 2.0;
        p.ok = in_b < 2.0e9 && pick_config(a, &p.c, &p.g);
        it = cache.emplace(k, p).first;
    }
    return &it->second;
}

}  // namespace

bool conv_ws_supported(const ConvArgs& a) {
    const bool off = !options().ws;  // A/B switch (TRTX_CONV_NOWS)
    return !off && a.up_C == 0 && ws_plan(a)->ok;   // a folded upsample is the implicit-GEMM main kernel's
}

int32_t conv_ws_f16(const ConvArgs& a0, hipStream_t s) {
    WsPlan* pl = ws_plan(a0);
    if (!pl->ok) return TRTX_ERR_UNSUPPORTED;
    ConvArgs a = a0;
    const unsigned in_bytes = (unsigned)((((size_t)a0.N * a0.H * a0.W - 1) * a0.ld_in + a0.Cin) * 2);
    if (pl->c.taps == 1) {  // 1x1: one row of N*H*W pixels
        const int M = a0.N * a0.Ho * a0.Wo;
        a.N = 1; a.H = 1; a.W = M; a.Ho = 1; a.Wo = M;
    }
    const int32_t st = dispatch_ws(a, pl->c, pl->g, in_bytes, s, &pl->occ);
    if (st != TRTX_OK) return st;
    return check_launch("conv_ws_f16");
}

}  // namespace trtx
