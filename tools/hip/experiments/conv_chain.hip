// Fused convolution CHAINS for gfx950 (MI355X): two or three consecutive convolutions of the reference graphs run as ONE launch over
// a spatial tile, the intermediate tensors never leave the CU's LDS.
//
//   * C2f bottleneck   3x3 -> 3x3 (+ shortcut)            yolov8/src/block.cpp:98-110   (11 pairs in YOLOv8n)
//   * detect-head arm  3x3 -> 3x3 -> 1x1 (+ bias)          yolov8/src/model.cpp:188-251  (6 arms: cv2 / cv3 x 3 levels)
//   * any single 3x3 stride-1 convolution (a one-stage chain): the "patch" form of the implicit GEMM
//
// Why (profiles/r02_layer_table.txt): every launch of the implicit-GEMM kernel pays ~6.5 us of fixed cost and a
// 3x3 layer drags every input pixel through the L1 -> LDS path nine times (the k-loop of conv_igemm.hip is bound by exactly that
// path).  Here a workgroup owns a TH x TW output tile of one image:
//
//   1. the input patch (tile + halo, all channels) is brought L2 -> LDS ONCE by buffer_load ... lds; pixels outside the image and
//      channels beyond Cin are the buffer descriptor's zero fill (= the convolution's zero padding);
//   2. stage s is a GEMM  [region pixels] x [Cout] x [taps * Cin]:  the A fragments are read straight out of the resident patch
//      (ds_read_b128 at a per-lane base + a wave-uniform tap / channel-slice offset).  The WEIGHTS either sit in LDS for the whole
//      launch (small-channel chains: all stages' weights are loaded next to the patch, the k-loops run without a single barrier or
//      wait) or stream through two LDS slots of `ks` 32-wide k-steps each (LDS-DMA of slot j+1 under the MFMAs of slot j, one
//      s_waitcnt + s_barrier per slot).  Phase stamps on MI355X (tools/chain_stamps.py) showed what a barrier per 32-wide step
//      costs: ~900 cycles per step for 240 cycles of MFMA work;
//   3. the epilogue (bias, activation, fp16 rounding, shortcut) writes the stage's result into the next patch in LDS; positions
//      outside the image are written as zeros, because they are the NEXT convolution's padding;
//   4. the last patch is copied out with 16-byte stores into the (possibly strided) NHWC slice.
//
// A 3x3 stage computes its consumers' halo too: stage regions shrink by one ring per following 3x3 (8x16 tile, two 3x3:
// 10x18 -> 8x16, 1.4x the MFMA work of the first stage) - bought back many times over by not writing and re-reading the
// intermediate tensors and by two launches fewer.
//
// LDS layout of a patch: planes of 32 channels, 64 bytes per pixel per plane, row pitch a multiple of 4 pixels; the four 16-byte
// chunks of a pixel are XOR-swizzled with key(x) = 2 * ((x >> 2) & 1), x = column inside the patch.  Brute-forced over the
// ds_read_b128 lane groups of MI355X_MICROARCH.md: conflict-free for any run of 16 consecutive pixels at ANY alignment (so for
// every tap shift), 2-way where a fragment wraps to the next patch row.  The key does not depend on the row: a tap's row offset
// is a wave-uniform constant, a tap's column offset selects one of three precomputed per-lane bases.
//
// Arithmetic = conv_igemm.hip's: fp16 operands, fp32 accumulation on v_mfma_f32_16x16x32_f16 in the same K order (tap, channel),
// bias added in fp32, ONE rounding to fp16 after the activation, shortcut added to the rounded value and rounded again.
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

// Ablation switches for the timing experiments (tools/chain_bench.py; build with -DTRTX_CHAIN_ABLATE to get them from the TRTX_CHAIN_DBG
// environment variable).  In the product build the flag word is the constant 0 and every test on it folds away.
#ifdef TRTX_CHAIN_ABLATE
#define CHAIN_DBG(flags) (flags)
#else
#define CHAIN_DBG(flags) 0
#endif

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr unsigned kOOB = 0x80000000u;  // beyond any num_records: the buffer load returns 0 and touches no memory
constexpr int kMaxStages = 3;

// one stage as the kernel sees it (every field wave-uniform)
struct StageArgs {
    const void* wgt;     // fp16 [16 * NFRAG][kpad]
    const float* bias;   // [16 * NFRAG]
    unsigned w_bytes;
    int taps;            // 3: 3x3 (pad 1), 1: 1x1
    int ks;              // streamed weights: 32-wide k-steps per ring slot
    int w_lds;           // resident weights: LDS offset of this stage's image [nk][Cout][32]
    int kc;              // 32-channel planes of the source patch
    int kpad;            // halfs per weight row = taps * taps * kc * 32
    int act;
    float alpha;
    int residual;        // add the chain input (centre pixels) after the activation
    int zero_outside;    // region pixels outside the image are stored as zeros (they are the next 3x3's padding)
    int rh, rw, npix, nfr;  // region, its pixel count, its 16-pixel fragments
    float inv_rw;
    int src_off, src_pitch, src_plane, src_y0, src_x0;  // patch read: pixel of region (0,0)'s tap (0,0)
    int dst_off, dst_pitch, dst_plane;                  // patch written (region pixel (y,x) -> patch pixel (y,x))
    int img_y0, img_x0;                                 // image coordinates of region (0,0) relative to the tile origin
};

struct ChainArgs {
    const void* in;
    void* out;
    unsigned in_bytes;
    int N, H, W, Cin, ld_in, Cout, ld_out;
    int TH, TW, tiles_x, tiles_per_img, total_tiles, xcd_chunk;
    float inv_tpi, inv_tx;
    int halo;                                   // input patch origin = tile origin - halo
    int in_off, in_ph, in_pw, in_pitch, in_plane, in_kc, in_groups;  // groups = 16-pixel DMA groups per plane
    float inv_in_pitch;
    int res_y0, res_x0;                         // chain input of region pixel (0,0) of a residual stage, relative to its img origin
    int out_off, out_pitch, out_plane;          // the last stage's patch
    float inv_tw, inv_cpp;
    int bias_off;       // [kMaxStages][128] fp32: the stages' bias vectors, DMA'd next to the patch
    int ring_off;       // streamed weights: two slots of slot_bytes each
    int slot_bytes;
    int nstages;
    unsigned long long* stamps;  // timing experiments: 16 s_memtime stamps per workgroup (wave 0), or nullptr
    int dbg;  // timing experiments (TRTX_CHAIN_DBG): 1 no MFMA, 2 no LDS fragment reads, 4 no weight DMA, 8 no epilogue math, 16 no patch load, 32 no copy-out
    StageArgs st[kMaxStages];
};

// keeps a wave-uniform value in an SGPR: without it the compiler re-loads stage fields from the kernel-argument segment inside the
// k-loop (s_load + s_waitcnt lgkmcnt(0), which also drains the LDS fragment reads of that step)
__device__ __forceinline__ int pin(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

// L2 -> LDS DMA (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes land lane-linear at the LDS address in M0), issued as inline assembly
// ON PURPOSE.  Through the builtin the compiler knows the instruction writes LDS, cannot prove that the destination (a ring slot at a
// run-time offset) differs from what the following ds_reads touch, and orders them with s_waitcnt vmcnt(..) in front of every fragment
// read: the DMA of the NEXT slot then has to land before THIS slot's MFMAs may start.  The kernel orders DMA against reads itself: one
// s_waitcnt vmcnt(0) + s_barrier where a slot (or the patch) is first read.  (Tried instead: global_load -> registers -> ds_write_b128.
// Measured slower on MI355X at this kernel's one or two workgroups per CU - weights 5.4k instead of 3.1k cycles per 18-step stage,
// patch load 7.5k instead of 3.8k - so the DMA stays, at ~170 cycles of issue per instruction.)
typedef int intx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ intx4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return intx4{__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull)), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull)),
                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
__device__ __forceinline__ void dma16(const intx4& rsrc, const char* lds_dst, unsigned voff) {
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_ptr_t) const_cast<char*>(lds_dst));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}

__device__ __forceinline__ int swz_w(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }  // weight rows: conv_igemm.hip's swz<32>
__device__ __forceinline__ int key_x(int x) { return ((x >> 2) & 1) << 1; }                         // patch pixels (header comment)

// exact x / d for small non-negative x with a float reciprocal estimate fixed up by one step
__device__ __forceinline__ int div_small(int x, int d, float inv) {
    int q = (int)((float)x * inv);
    int r = x - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; }
    return q;
}

__device__ __attribute__((noinline)) float chain_act_rare(float v, int act, float alpha) {
    if (act == ACT_LEAKY) return v > 0.f ? v : v * alpha;
    if (act == ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (act == ACT_TANH) return tanhf(v);
    if (act == ACT_MISH) return mish_ref(v);
    return v;
}

// NFRAG: 16-channel output fragments (Cout = 16 * NFRAG, the same for every stage of a chain); MI: region fragments a wave may own
// (fragment f of a region belongs to wave f % 4); RES: every stage's weights are resident in LDS (no ring, no barriers in the k-loops).
template <int NFRAG, int MI, bool RES>
__global__ __launch_bounds__(256) void conv_chain_f16_kernel(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int SB = 16 * NFRAG * 64;  // bytes of one 32-wide k-step of weights: all Cout rows x 32 halfs

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;
    if (p.xcd_chunk) {  // XCD-aware order (conv_igemm.hip): each XCD walks a contiguous range of tiles, halos stay in one L2
        tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
        if (tile >= p.total_tiles) return;
    }
    const int n = div_small(tile, p.tiles_per_img, p.inv_tpi);
    const int rt = tile - n * p.tiles_per_img;
    const int ty = div_small(rt, p.tiles_x, p.inv_tx);
    const int tx = rt - ty * p.tiles_x;
    const int y0 = ty * p.TH, x0 = tx * p.TW;

    int stamp_i = 0;
    auto stamp = [&]() {
        if (p.stamps && tid == 0 && tile < 512 && stamp_i < 16) p.stamps[tile * 16 + stamp_i] = __builtin_readcyclecounter();
        ++stamp_i;
    };
    stamp();  // 0: start
    const int dbg = CHAIN_DBG(p.dbg);
    const int g = lane >> 4;      // k-group of an MFMA operand / channel quad of an MFMA result
    const int lrow = lane & 15;   // pixel row of a fragment

    // ---- weights.  A 32-wide k-step of a stage is a tile [Cout rows][64 B]; a wave-instruction of the DMA fills 16 rows; this wave
    // owns the 16-row groups wave, wave + 4 of every step (NFRAG <= 8: at most two).  Row r's four 16-byte chunks are swizzled by
    // swz_w(r) on the source side (the DMA writes lane-linear), read back by the fragment loads with the same key.
    const int nbw = (NFRAG - wave + 3) / 4;
    const int brow = lane >> 2, blog = (lane & 3) ^ swz_w(brow);
    const int fb_off = lrow * 64 + ((g ^ swz_w(lrow)) << 4);
    intx4 rs_w = make_rsrc(p.st[0].wgt, p.st[0].w_bytes);
    unsigned b_goff[2] = {0, 0};
    int nk = 0;
    auto stage_weights = [&](const StageArgs& S) {
        rs_w = make_rsrc(S.wgt, S.w_bytes);
        nk = S.taps * S.taps * S.kc;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) b_goff[jj] = (unsigned)((((wave + 4 * jj) * 16 + brow) * S.kpad + blog * 8) * 2);
    };
    // k-steps [first, first + count) -> LDS at dst (step h at dst + h * SB)
    auto issue_steps = [&](int first, int count, char* dst) {
        if (dbg & 4) return;
        for (int h = 0; h < count; ++h) {
            if (first + h >= nk) break;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if (jj >= nbw) continue;
                dma16(rs_w, dst + h * SB + (wave + 4 * jj) * 1024, b_goff[jj] + (unsigned)(first + h) * 64u);
            }
        }
    };

    // ---- 0. weights first (all stages when resident, the first slot otherwise), then the bias vectors and the input patch: L2 -> LDS
    if (RES) {
        for (int s = 0; s < p.nstages; ++s) {
            stage_weights(p.st[s]);
            issue_steps(0, nk, smem + p.st[s].w_lds);
        }
    } else {
        stage_weights(p.st[0]);
        issue_steps(0, p.st[0].ks, smem + p.ring_off);
    }
    // the stages' bias vectors (<= 128 floats each) into LDS, one DMA per stage: a global load in the epilogue would cost a memory latency
    // per stage, and a compiler-visible global load makes hipcc put s_waitcnt vmcnt(..) in front of LDS reads inside the k-loop
    if (wave < p.nstages) {
        const intx4 rs_b = make_rsrc(p.st[wave].bias, (unsigned)(16 * NFRAG * 4));
        dma16(rs_b, smem + p.bias_off + wave * 1024, lane < 4 * NFRAG ? (unsigned)lane * 16u : kOOB);
    }
    {
        const intx4 rs_in = make_rsrc(p.in, p.in_bytes);
        const int img_row0 = n * p.H;
        for (int jj = wave; jj < p.in_groups; jj += 4) {
            const int pp = jj * 16 + (lane >> 2);
            const int py = div_small(pp, p.in_pitch, p.inv_in_pitch);
            const int px = pp - py * p.in_pitch;
            const int clog = (lane & 3) ^ key_x(px);  // this lane fills physical chunk (lane & 3) of pixel pp
            const int iy = y0 - p.halo + py, ix = x0 - p.halo + px;
            const bool ok = py < p.in_ph && px < p.in_pw && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && !(dbg & 16);
            const unsigned pix = (unsigned)(((img_row0 + iy) * p.W + ix) * p.ld_in + clog * 8) * 2u;
            for (int kc = 0; kc < p.in_kc; ++kc) {
                const bool okc = ok && (kc * 32 + clog * 8 < p.Cin);
                dma16(rs_in, smem + p.in_off + kc * p.in_plane + jj * 1024, okc ? pix + (unsigned)kc * 64u : kOOB);
            }
        }
    }
    stamp();  // 1: patch + first weights issued
    if (RES) {  // everything the launch reads from memory is in flight: one wait, one barrier, then no synchronisation until the epilogue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp();  // 2: patch (and weights) landed
    }

    floatx4 acc[MI][NFRAG];

    for (int s = 0; s < p.nstages; ++s) {
        const StageArgs& S = p.st[s];
        if (!RES && s > 0) {
            stage_weights(S);
            issue_steps(0, S.ks, smem + p.ring_off);
        }
        if (RES) nk = S.taps * S.taps * S.kc;
        const int s_taps = pin(S.taps), s_kc = pin(S.kc), s_ks = pin(S.ks), s_nfr = pin(S.nfr);
        const int s_src_plane = pin(S.src_plane), s_src_rowb = pin(S.src_pitch * 64), s_wlds = pin(S.w_lds);
        const int s_ring = pin(p.ring_off), s_slot = pin(p.slot_bytes);
        // ---- per-lane geometry of this wave's fragments: region pixel m = 16 f + lrow -> (y, x)
        int py[MI], px[MI];
        unsigned aq0[MI], aq1[MI], aq2[MI];  // fragment read address for filter column 0 / 1 / 2 (row and plane offsets are wave-uniform adds)
        bool pvalid[MI], factive[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            factive[i] = wave + 4 * i < s_nfr;  // wave-uniform: fragment f of a region belongs to wave f % 4
            const int m = (wave + 4 * i) * 16 + lrow;
            pvalid[i] = m < S.npix;
            const int mm = pvalid[i] ? m : 0;
            py[i] = div_small(mm, S.rw, S.inv_rw);
            px[i] = mm - py[i] * S.rw;
            // column q reads patch pixel x + q: 64 bytes further, and the chunk swizzle flips where x + q crosses a multiple of 4
            const int xp = S.src_x0 + px[i];
            const int a0 = S.src_off + ((S.src_y0 + py[i]) * S.src_pitch + xp) * 64;
            aq0[i] = (unsigned)(a0 + ((g ^ key_x(xp)) << 4));
            aq1[i] = (unsigned)(a0 + 64 + ((g ^ key_x(xp + 1)) << 4));
            aq2[i] = (unsigned)(a0 + 128 + ((g ^ key_x(xp + 2)) << 4));
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        }

        // ---- k-loop over the 32-wide steps in the order (filter row r, filter column q, channel plane kc).  What the phase stamps and
        // the ablation switches showed on MI355X (tools/chain_stamps.py): at one or two workgroups per CU this loop is bound by
        // INSTRUCTION ISSUE - a step's MFMAs are 130-250 cycles, a generic cursor (which tap? which plane? which slot? is this fragment
        // mine?) was ~500 cycles of scalar bookkeeping and branches per step, and removing the LDS fragment reads changed nothing.  So
        // the body is specialised until a step is: one address add per fragment, the fragment reads, the MFMAs.
        //   * the filter column is unrolled (three copies of the plane loop, each with its own precomputed per-lane base);
        //   * the number of fragments a wave works on is a compile-time constant FR = the count of the busiest wave (a wave that owns
        //     one fewer computes a throw-away fragment instead of branching around it);
        //   * streamed weights: step t lives in ring slot (t / ks) & 1; entering a slot = wait for its DMA (vmcnt(0)), barrier (every
        //     wave has left the slot that is refilled next - its fragment reads have been consumed by MFMAs), issue the DMA of the
        //     following slot.  Nothing else synchronises.
        int t = 0, sub = 0, par = 0;
        auto slot_sync = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t == 0 && s == 0 && p.stamps) stamp();  // 2: patch landed, first barrier passed
            issue_steps(t + s_ks, s_ks, smem + s_ring + (par ^ 1) * s_slot);
        };
        auto kloop = [&](auto frc) {
            constexpr int FR = decltype(frc)::value;
            auto step = [&](const unsigned (&aq)[MI], unsigned soff) {
                const char* sb;
                if (RES) {
                    sb = smem + s_wlds + t * SB + fb_off;
                } else {
                    if (sub == 0) slot_sync();
                    sb = smem + s_ring + par * s_slot + sub * SB + fb_off;
                }
                half8 bf[NFRAG], af[FR];
                if (!(dbg & 2)) {
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) bf[j] = *reinterpret_cast<const half8*>(sb + j * 1024);
#pragma unroll
                    for (int i = 0; i < FR; ++i) af[i] = *reinterpret_cast<const half8*>(smem + aq[i] + soff);
                } else {
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) bf[j] = half8{1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
                    for (int i = 0; i < FR; ++i) af[i] = half8{1, 1, 1, 1, 1, 1, 1, 1};
                }
                if (!(dbg & 1)) {
#pragma unroll
                    for (int i = 0; i < FR; ++i)
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
                }
                ++t;
                if (!RES) {
                    ++sub;
                    if (sub == s_ks) {
                        sub = 0;
                        par ^= 1;
                    }
                }
            };
            for (int r = 0; r < s_taps; ++r) {
                unsigned soff = (unsigned)(r * s_src_rowb);
                for (int kc = 0; kc < s_kc; ++kc, soff += (unsigned)s_src_plane) step(aq0, soff);
                if (s_taps > 1) {
                    soff = (unsigned)(r * s_src_rowb);
                    for (int kc = 0; kc < s_kc; ++kc, soff += (unsigned)s_src_plane) step(aq1, soff);
                    soff = (unsigned)(r * s_src_rowb);
                    for (int kc = 0; kc < s_kc; ++kc, soff += (unsigned)s_src_plane) step(aq2, soff);
                }
            }
        };
        {
            const int nf_hi = (s_nfr + 3) >> 2;  // fragments of the busiest wave
            if (MI >= 6 && nf_hi > 3) {
                if (nf_hi == 6) kloop(std::integral_constant<int, (MI >= 6 ? 6 : MI)>{});
                else if (nf_hi == 5) kloop(std::integral_constant<int, (MI >= 5 ? 5 : MI)>{});
                else kloop(std::integral_constant<int, (MI >= 4 ? 4 : MI)>{});
            } else if (MI >= 3 && nf_hi >= 3) {
                kloop(std::integral_constant<int, (MI >= 3 ? 3 : MI)>{});
            } else if (MI >= 2 && nf_hi == 2) {
                kloop(std::integral_constant<int, (MI >= 2 ? 2 : MI)>{});
            } else {
                kloop(std::integral_constant<int, 1>{});
            }
        }
        stamp();  // 3 + 3 s: k-loop of stage s done

        // ---- epilogue: bias, activation, fp16, (+ chain input), into the destination patch; lane = 4 channels of one pixel.  The
        // activation is chosen once (wave-uniform), the bias vector is read once: what sits inside the per-fragment loops is arithmetic
        float4 bias_r[NFRAG];
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) bias_r[j] = *reinterpret_cast<const float4*>(smem + p.bias_off + s * 1024 + (j * 16 + g * 4) * 4);
        const int s_dst_plane = pin(S.dst_plane), s_dst_off = pin(S.dst_off), s_in_plane = pin(p.in_plane);
        const bool s_res = S.residual != 0, s_zero = S.zero_outside != 0;
        auto epilogue = [&](auto actc) {
            constexpr int ACT = decltype(actc)::value;  // ACT_SILU / ACT_RELU / ACT_NONE, or -1: by the run-time code
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (!factive[i]) continue;
                const int y = py[i], x = px[i];
                const int iy = y0 + S.img_y0 + y, ix = x0 + S.img_x0 + x;
                const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const int kd = key_x(x) << 4;
                const int dst0 = s_dst_off + (y * S.dst_pitch + x) * 64 + (g & 1) * 8;
                // chain input of this pixel (shortcut): input patch pixel (img_y0 + y + halo, img_x0 + x + halo)
                const int rx = S.img_x0 + x + p.halo;
                const int kr = key_x(rx) << 4;
                const int res0 = p.in_off + ((S.img_y0 + y + p.halo) * p.in_pitch + rx) * 64 + (g & 1) * 8;
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) {
                    const float4 b = bias_r[j];
                    float v[4] = {acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z, acc[i][j][3] + b.w};
                    half4 h;
                    if (dbg & 8) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = round_to_half(v[e]);
                    } else if (ACT == ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = round_to_half(v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[e])));
                    } else if (ACT == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = round_to_half(v[e] > 0.f ? v[e] : 0.f);
                    } else if (ACT == ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = round_to_half(v[e]);
                    } else {
#pragma nounroll
                        for (int e = 0; e < 4; ++e) h[e] = round_to_half(chain_act_rare(v[e], S.act, S.alpha));
                    }
                    const int chunk = ((j & 1) * 2 + (g >> 1)) << 4;  // 16-byte chunk of the 32-channel plane j >> 1
                    if (s_res) {
                        const half4 rv = *reinterpret_cast<const half4*>(smem + res0 + (j >> 1) * s_in_plane + (chunk ^ kr));
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = round_to_half((float)h[e] + (float)rv[e]);
                    }
                    if (s_zero && !inside) h = half4{0, 0, 0, 0};
                    if (pvalid[i]) {
                        *reinterpret_cast<half4*>(smem + dst0 + (j >> 1) * s_dst_plane + (chunk ^ kd)) = h;
                        // Cout % 32 == 16: channels [Cout, Cout + 16) of the last plane belong to nobody, but the next stage's k-step reads the
                        // whole plane (against zero weights) - and 0 x (whatever LDS held) must not be NaN
                        if ((NFRAG & 1) && j == NFRAG - 1) *reinterpret_cast<half4*>(smem + dst0 + (j >> 1) * s_dst_plane + ((chunk + 32) ^ kd)) = half4{0, 0, 0, 0};
                    }
                }
            }
        };
        if (S.act == ACT_SILU) epilogue(std::integral_constant<int, ACT_SILU>{});
        else if (S.act == ACT_NONE) epilogue(std::integral_constant<int, ACT_NONE>{});
        else if (S.act == ACT_RELU) epilogue(std::integral_constant<int, ACT_RELU>{});
        else epilogue(std::integral_constant<int, -1>{});
        stamp();  // 4 + 3 s: epilogue done
        __syncthreads();  // the patch is complete (and every wave is done with the ring) before the next stage / the copy-out reads it
        stamp();  // 5 + 3 s: every wave past the barrier
    }

    // ---- copy-out: last patch -> NHWC slice, one lane = 8 channels of one pixel (16-byte loads and stores)
    {
        const int cpp = p.Cout >> 3;
        const int total = p.TH * p.TW * cpp;
        _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
        for (int idx = tid; idx < total && !(dbg & 32); idx += 256) {
            const int pix = div_small(idx, cpp, p.inv_cpp);
            const int c = idx - pix * cpp;
            const int y = div_small(pix, p.TW, p.inv_tw);
            const int x = pix - y * p.TW;
            const int iy = y0 + y, ix = x0 + x;
            if (iy >= p.H || ix >= p.W) continue;
            const half8 v = *reinterpret_cast<const half8*>(smem + p.out_off + (c >> 2) * p.out_plane + (y * p.out_pitch + x) * 64 + (((c & 3) ^ key_x(x)) << 4));
            *reinterpret_cast<half8*>(out + ((size_t)(n * p.H + iy) * p.W + ix) * p.ld_out + c * 8) = v;
        }
    }
    stamp();  // last: copy-out issued
}

// test support: fills every CU's LDS with fp16 NaN patterns.  LDS is not cleared between kernels, so a kernel that reads LDS bytes it
// never wrote sees whatever the previous kernel left; the parity tests poison it first (tests/test_gpu_conv_chain.py).
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_words[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_words[i] = 0x7fff7fffu;
    __syncthreads();
    if (lds_words[(threadIdx.x * 37) % (160 * 1024 / 4)] == 1u) *sink = 1u;  // keeps the stores alive
}

// ------------------------------------------------------------------------------------------------------------------------------
// host side: geometry, LDS plan, dispatch

struct TileChoice {
    int th, tw;
};

int round_up(int v, int a) { return (v + a - 1) / a * a; }

// fills `a` for tile (th, tw); ks = 32-wide k-steps per weight-ring slot, 0 = all weights resident in LDS.  Returns the dynamic LDS
// bytes needed, or 0 when the tile cannot be used
size_t plan_chain(const ChainDesc& d, int th, int tw, int ks, ChainArgs* a, int* mi_needed) {
    const int ns = d.nstages;
    int halo = 0;
    for (int s = 0; s < ns; ++s) halo += d.st[s].k == 3 ? 1 : 0;
    ChainArgs& A = *a;
    memset(&A, 0, sizeof(A));
    A.in = d.in;
    A.out = d.out;
    A.N = d.N; A.H = d.H; A.W = d.W; A.Cin = d.Cin; A.ld_in = d.ld_in; A.ld_out = d.ld_out;
    A.Cout = d.st[ns - 1].cout;
    A.in_bytes = (unsigned)((((size_t)d.N * d.H * d.W - 1) * d.ld_in + d.Cin) * 2);
    A.TH = th; A.TW = tw;
    A.tiles_x = (d.W + tw - 1) / tw;
    const int tiles_y = (d.H + th - 1) / th;
    A.tiles_per_img = A.tiles_x * tiles_y;
    A.total_tiles = A.tiles_per_img * d.N;
    A.xcd_chunk = (A.total_tiles + 7) / 8;
    A.inv_tpi = 1.0f / (float)A.tiles_per_img;
    A.inv_tx = 1.0f / (float)A.tiles_x;
    A.halo = halo;
    A.nstages = ns;
    // input patch
    A.in_ph = th + 2 * halo;
    A.in_pw = tw + 2 * halo;
    A.in_pitch = round_up(A.in_pw, 4);
    A.in_kc = (d.Cin + 31) / 32;
    A.in_groups = (A.in_ph * A.in_pitch + 15) / 16;
    A.in_plane = A.in_groups * 1024;
    A.inv_in_pitch = 1.0f / (float)A.in_pitch;
    A.inv_tw = 1.0f / (float)tw;
    A.inv_cpp = 1.0f / (float)(A.Cout / 8);
    // buffers: [0] = input patch, then one per stage; a stage's output may reuse the buffer of the patch its PRODUCER read, unless
    // that is the chain input and a later (or this) stage still adds it as shortcut
    struct Buf { size_t off, bytes; };
    Buf bufs[1 + kMaxStages];
    size_t top = 0;
    bufs[0] = {0, (size_t)A.in_kc * A.in_plane};
    top = bufs[0].bytes;
    int last_res = -1;
    for (int s = 0; s < ns; ++s)
        if (d.st[s].residual) last_res = s;
    int max_frags = 0;
    int rem = halo;  // rings the region of stage s carries beyond the tile
    for (int s = 0; s < ns; ++s) {
        const ChainStageDesc& D = d.st[s];
        StageArgs& S = A.st[s];
        if (D.k == 3) --rem;
        const int cin = s == 0 ? d.Cin : d.st[s - 1].cout;
        S.wgt = D.wgt;
        S.bias = D.bias;
        S.taps = D.k;
        S.kc = (cin + 31) / 32;
        S.kpad = D.k * D.k * S.kc * 32;
        S.w_bytes = (unsigned)((size_t)D.cout * S.kpad * 2);
        S.act = D.act;
        S.alpha = D.alpha;
        S.residual = D.residual;
        S.zero_outside = s + 1 < ns ? 1 : 0;
        S.rh = th + 2 * rem;
        S.rw = tw + 2 * rem;
        S.npix = S.rh * S.rw;
        S.nfr = (S.npix + 15) / 16;
        S.inv_rw = 1.0f / (float)S.rw;
        max_frags = std::max(max_frags, S.nfr);
        S.img_y0 = -rem;
        S.img_x0 = -rem;
        // source patch: region (0,0)'s tap (0,0).  The source holds the region of the previous stage (or the input patch), whose
        // ring count is rem + (k == 3): for a 3x3 the window starts one pixel up / left of the output pixel = source pixel (y, x);
        // for a 1x1 source and region coincide.
        if (s == 0) {
            S.src_off = 0;
            S.src_pitch = A.in_pitch;
            S.src_plane = A.in_plane;
        } else {
            S.src_off = A.st[s - 1].dst_off;
            S.src_pitch = A.st[s - 1].dst_pitch;
            S.src_plane = A.st[s - 1].dst_plane;
        }
        S.src_y0 = 0;
        S.src_x0 = 0;
        S.dst_pitch = round_up(S.rw, 4);
        const int dst_groups = (S.rh * S.dst_pitch + 15) / 16;
        S.dst_plane = dst_groups * 1024;
        const size_t need = (size_t)((D.cout + 31) / 32) * S.dst_plane;
        // reuse: the buffer read by stage s - 1 (index s - 1 in bufs; bufs[0] is the input) is dead once stage s - 1 is done
        int reuse = -1;
        if (s >= 1) {
            const int cand = s - 1;
            const bool input_still_needed = cand == 0 && last_res >= s;
            if (!input_still_needed && bufs[cand].bytes >= need) reuse = cand;
        }
        if (reuse >= 0) {
            bufs[s + 1] = {bufs[reuse].off, need};
        } else {
            bufs[s + 1] = {top, need};
            top += round_up((int)need, 1024);
        }
        S.dst_off = (int)bufs[s + 1].off;
    }
    A.in_off = 0;
    A.out_off = A.st[ns - 1].dst_off;
    A.out_pitch = A.st[ns - 1].dst_pitch;
    A.out_plane = A.st[ns - 1].dst_plane;
    A.bias_off = (int)top;
    top += kMaxStages * 1024;
    const int SB = A.Cout * 64;  // one 32-wide k-step of weights
    if (ks == 0) {  // resident: one image per stage
        for (int s = 0; s < ns; ++s) {
            A.st[s].w_lds = (int)top;
            A.st[s].ks = A.st[s].taps * A.st[s].taps * A.st[s].kc;
            top += (size_t)A.st[s].ks * SB;
        }
        A.ring_off = 0;
        A.slot_bytes = 0;
    } else {
        int ks_max = 1;
        for (int s = 0; s < ns; ++s) {
            A.st[s].ks = std::min(ks, A.st[s].taps * A.st[s].taps * A.st[s].kc);
            ks_max = std::max(ks_max, A.st[s].ks);
        }
        A.ring_off = (int)top;
        A.slot_bytes = ks_max * SB;
        top += 2 * (size_t)A.slot_bytes;
    }
    *mi_needed = (max_frags + 3) / 4;
    return top;
}

template <int NFRAG, int MI, bool RES>
int32_t launch_chain(const ChainArgs& a, size_t lds, hipStream_t s) {
    static bool attr_done[16] = {};  // per device: dynamic LDS above the default limit needs the attribute once
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_chain_f16_kernel<NFRAG, MI, RES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_done[dev] = true;
    }
    TRTX_LAUNCH((conv_chain_f16_kernel<NFRAG, MI, RES>), dim3(a.xcd_chunk * 8), dim3(256), lds, s, a);
    return TRTX_OK;
}

template <int NFRAG, bool RES>
int32_t launch_mi(const ChainArgs& a, int mi, size_t lds, hipStream_t s) {
    if (mi <= 1) return launch_chain<NFRAG, 1, RES>(a, lds, s);
    if (mi == 2) return launch_chain<NFRAG, 2, RES>(a, lds, s);
    if (mi == 3) return launch_chain<NFRAG, 3, RES>(a, lds, s);
    if constexpr (NFRAG <= 2) {
        if (mi <= 6) return launch_chain<NFRAG, 6, RES>(a, lds, s);
    }
    return TRTX_ERR_UNSUPPORTED;
}

template <bool RES>
int32_t launch_nf(const ChainArgs& a, int nfrag, int mi, size_t lds, hipStream_t s) {
    switch (nfrag) {
        case 1: return launch_mi<1, RES>(a, mi, lds, s);
        case 2: return launch_mi<2, RES>(a, mi, lds, s);
        case 4: return launch_mi<4, RES>(a, mi, lds, s);
        case 5: return launch_mi<5, RES>(a, mi, lds, s);
        case 8: return launch_mi<8, RES>(a, mi, lds, s);
        default: return TRTX_ERR_UNSUPPORTED;
    }
}

bool desc_ok(const ChainDesc& d) {
    if (d.nstages < 1 || d.nstages > kMaxStages || d.N < 1 || d.H < 1 || d.W < 1) return false;
    if (d.Cin % 8 || d.ld_in % 8 || d.ld_out % 8 || d.Cin > 512) return false;
    const int cout = d.st[0].cout;
    if (cout != 16 && cout != 32 && cout != 64 && cout != 80 && cout != 128) return false;
    for (int s = 0; s < d.nstages; ++s) {
        const ChainStageDesc& D = d.st[s];
        if (D.cout != cout || (D.k != 1 && D.k != 3)) return false;
        if (D.residual && d.Cin != cout) return false;
    }
    if (((size_t)d.N * d.H * d.W * d.ld_in) * 2 >= 2000000000ull) return false;  // 32-bit buffer offsets
    return true;
}

// tile and weight mode by a small cost model, in units of one MFMA issue slot (16 cycles) of a wave; the constants come from the
// phase stamps of tools/chain_stamps.py on MI355X.  A workgroup costs: per 32-wide k-step the MFMAs of its busiest wave
// (fragments x column fragments) + ~8 for the fragment loads that do not overlap them; per weight-ring slot ~30 for the wait +
// barrier + DMA issue (nothing when the weights are resident); per stage ~60 for the epilogue; ~250 fixed (patch load, copy-out,
// launch ramp).  The 256 CUs work the tiles off in parallel; co-resident workgroups share the MFMA pipes, so a second one only
// hides the first one's stalls: a LONE workgroup per CU is priced 1.3x.  Ties go to the larger tile.
bool choose(const ChainDesc& d, ChainArgs* a, int* mi, int* ks_out, size_t* lds) {
    static const TileChoice cands[] = {{16, 16}, {8, 16}, {8, 8}, {4, 8}, {4, 4}};
    static const int modes[] = {0, 4, 2, 1};  // 0 = resident weights; n = k-steps per ring slot
    int fth = d.tile_h, ftw = d.tile_w, fks = -1;
    if (const char* e = getenv("TRTX_CHAIN_TILE")) {  // A/B experiments: "8x16"
        int h = 0, w = 0;
        if (sscanf(e, "%dx%d", &h, &w) == 2 && h > 0 && w > 0) { fth = h; ftw = w; }
    }
    if (const char* e = getenv("TRTX_CHAIN_KS")) fks = atoi(e);  // A/B experiments: k-steps per ring slot (0 = resident)
    bool found = false;
    double best_cost = 0;
    for (int c = 0; c < (int)(sizeof(cands) / sizeof(cands[0])); ++c) {
        int th = cands[c].th, tw = cands[c].tw;
        if (fth > 0) { th = fth; tw = ftw; }
        for (int mode : modes) {
            if (fks >= 0 && mode != fks) continue;
            ChainArgs t;
            int m = 0;
            const size_t need = plan_chain(d, th, tw, mode, &t, &m);
            if (need == 0 || need > 160 * 1024 || m > 6 || (m > 3 && t.Cout > 32)) continue;  // 6 fragments per wave: only with <= 2 column fragments (registers)
            const int per_cu = std::max(1, std::min(4, (int)(160 * 1024 / need)));
            double tile_cost = 250.0;
            for (int s = 0; s < t.nstages; ++s) {
                const int nk = t.st[s].taps * t.st[s].taps * t.st[s].kc;
                tile_cost += (double)nk * ((double)((t.st[s].nfr + 3) / 4) * (t.Cout / 16) + 8.0) + 60.0;
                if (mode != 0) tile_cost += 30.0 * ((nk + t.st[s].ks - 1) / t.st[s].ks);
            }
            const double cost = std::max(1.0, (double)t.total_tiles / 256.0) * tile_cost * (per_cu == 1 ? 1.3 : 1.0);
            if (!found || cost < best_cost * 0.999) {
                found = true;
                best_cost = cost;
                *a = t;
                *mi = m;
                *lds = need;
                *ks_out = mode;
            }
        }
        if (fth > 0) break;
    }
    return found;
}

}  // namespace

int32_t conv_chain_poison_lds(unsigned* device_word, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_done = true;
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 160 * 1024, s, device_word);  // one workgroup per CU at a time, four rounds
    return check_launch("poison_lds");
}

static unsigned long long* g_chain_stamps = nullptr;
void conv_chain_set_stamps(unsigned long long* device_buffer_512x16) { g_chain_stamps = device_buffer_512x16; }

size_t conv_chain_weight_halfs(int cin, int cout, int k) { return (size_t)cout * k * k * ((cin + 31) / 32) * 32; }

void conv_chain_pack_weights(const float* w_kcrs, int cout, int cin, int k, const float* ch_scale, uint16_t* packed) {
    const int kc = (cin + 31) / 32, kpad = k * k * kc * 32;
    memset(packed, 0, sizeof(uint16_t) * (size_t)cout * kpad);
    for (int co = 0; co < cout; ++co) {
        const float sc = ch_scale ? ch_scale[co] : 1.0f;
        for (int c = 0; c < cin; ++c)
            for (int r = 0; r < k; ++r)
                for (int q = 0; q < k; ++q) {
                    const _Float16 h = (_Float16)(w_kcrs[(((size_t)co * cin + c) * k + r) * k + q] * sc);
                    uint16_t u;
                    memcpy(&u, &h, 2);
                    packed[(size_t)co * kpad + (size_t)(r * k + q) * kc * 32 + c] = u;
                }
    }
}

bool conv_chain_supported(const ChainDesc& d) {
    if (!desc_ok(d)) return false;
    ChainArgs a;
    int mi = 0, nst = 0;
    size_t lds = 0;
    return choose(d, &a, &mi, &nst, &lds);
}

int32_t conv_chain_describe(const ChainDesc& d, int* th, int* tw, int* lds_bytes, int* nst) {
    if (!desc_ok(d)) return TRTX_ERR_UNSUPPORTED;
    ChainArgs a;
    int mi = 0, ring = 0;
    size_t lds = 0;
    if (!choose(d, &a, &mi, &ring, &lds)) return TRTX_ERR_UNSUPPORTED;
    if (th) *th = a.TH;
    if (tw) *tw = a.TW;
    if (lds_bytes) *lds_bytes = (int)lds;
    if (nst) *nst = ring;
    return TRTX_OK;
}

int32_t conv_chain_f16(const ChainDesc& d, hipStream_t s) {
    if (!desc_ok(d)) return TRTX_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(d.out) & 15) || (reinterpret_cast<uintptr_t>(d.in) & 15)) return TRTX_ERR_UNSUPPORTED;
    ChainArgs a;
    int mi = 0, nst = 0;
    size_t lds = 0;
    if (!choose(d, &a, &mi, &nst, &lds)) return TRTX_ERR_UNSUPPORTED;
    if (const char* e = getenv("TRTX_CHAIN_DBG")) a.dbg = atoi(e);
    a.stamps = g_chain_stamps;
    const int nfrag = a.Cout / 16;
    const int32_t st = nst == 0 ? launch_nf<true>(a, nfrag, mi, lds, s) : launch_nf<false>(a, nfrag, mi, lds, s);
    if (st != TRTX_OK) return st;
    return check_launch("conv_chain_f16");
}

}  // namespace trtx
