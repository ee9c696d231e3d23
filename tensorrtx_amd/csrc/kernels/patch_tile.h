// Device code of the resident-patch 3x3 convolution: the tile function and the one-problem kernel entry point, shared by conv_igemm.hip (fp16 operands,
// + the grouped entry point) and conv_igemm_f32.hip (fp32 operands, round 5).
#pragma once
#include "igemm_tile.h"

namespace trtx {
namespace {

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 pad-1 variant with a RESIDENT INPUT PATCH ("patch", tactic ConvArgs::t_ws == 3; a product tactic since round 5).  The kernels
// above bring an A tile from global memory to LDS for every (tile, tap, channel slice): a 3x3 layer's input crosses the global -> LDS path
// nine times per column tile, and on YOLOv8n / ResNet-50 that path - not HBM, not the MFMA pipe - is the largest term of a step
// (tools/lds_fill_model.py: 5.6 GB per b32 step for 2.3 GB of HBM bytes).  Here an output tile is a TH x 16 block of ONE image, its
// (TH + 2) x 18 input patch is brought to LDS once (one plane per 32-channel slice, conv_ws.hip's swizzled layout), the nine taps read
// their A fragments from it at shifted addresses, and only the weight tile of a k-step (BN x 32) streams through the three-stage ring:
// per 128 output pixels and 64 -> 64 channels 30 + 73 KB instead of 147 + 73 KB through the fill path (16 rows: 27 + 37 per 128 pixels).
// K is walked (tap, channel slice) as in the main kernel and every output element accumulates in the same order with the same MFMA: results
// are bit-identical to the main kernel's (tests/test_gpu_conv.py treats it as one more exchangeable tile shape, ConvArgs::t_ws == 3).
// All index arithmetic lives in patch_index.h and is replayed lane by lane on the CPU (tests/test_patch_index_cpu.py).
// Written at the end of round 4 without GPU minutes (CPU replay of its index arithmetic, ISA scan); first run in round 5 (profiles/r05_patch_shape_ab.txt,
// r05_patch_r3_first_run.txt): bit-identical to the main kernel on every shape on the first launch, 64 -> 64 @ 80x80 34 vs 39-44 us, 64 -> 80 42 vs 48-53,
// 80 -> 80 54 vs 70, 128 -> 64 @ 40x40 20 vs 22-25 (input flushed from the caches); in the whole step, where a layer's input is still warm from its producer,
// +0.3-1.3 % on the bench line.  A candidate of the tactic timing wherever patch_possible() holds, and the kernel of the grouped launches whose members all qualify.
template <int NFRAG, int KC, int MI>
constexpr int patch_lds_bytes() {
    return KC * (4 * MI + 2) * patchidx::kPitch * patchidx::kPixelBytes + 3 * ((16 * NFRAG + 63) / 64) * 64 * 64;
}
// one output tile (T) of one problem (p); `smem` is the workgroup's patch_lds_bytes<NFRAG, KC, MI>() of LDS
// F32 (round 5): fp32 operands exactly as igemm_tile.h's F32 - the launcher states the input geometry in 2-byte units, a plane is 16 floats of every patch
// pixel, a fragment read is four consecutive channels whose component s feeds the s-th v_mfma_f32_16x16x4_f32, every 16-channel step sums from zero into a
// partial that joins the running total in step order (the two-level sum), the total starts at the bias, 16-byte fp32 stores straight from the accumulators:
// the same bits as conv_igemm_tile<..., F32>.  For fp32 the fill path is what bounds the implicit-GEMM kernel (a 128 x 64 tile needs 12 KB of LDS fill per
// 262 kFLOP = 7.4 TB/s of L2 -> LDS traffic at the matrix pipe's peak; measured: the fill alone takes 0.8 of the MFMA time) - here 4 KB + the patch once.
template <int NFRAG, int KC, int MI, bool F32 = false>
__device__ __forceinline__ void conv_patch_tile(const ConvArgs& p, unsigned in_bytes, unsigned w_bytes, const patchidx::Tile& T, char* smem) {
    namespace px = patchidx;
    constexpr int BN = 16 * NFRAG;
    constexpr int TH = 4 * MI;
    constexpr int PLANE = (TH + 2) * px::kPitch * px::kPixelBytes;
    constexpr int PIECES = (TH + 2) * px::kPitch / 16;            // DMA pieces per plane
    constexpr int B_PASSES = (BN + 63) / 64;
    constexpr int STAGE_B = B_PASSES * 64 * 64;                    // weight tile of one k-step (rows beyond BN are dummy targets)
    constexpr int NST = 3;
    constexpr int PATCH_BYTES = KC * PLANE;
    constexpr int LDS_BYTES = PATCH_BYTES + NST * STAGE_B;
    constexpr int NK = 9 * KC;
    static_assert(MI == 2 || MI == 4, "8- or 16-row tiles");
    static_assert(PLANE % 1024 == 0, "a plane is a whole number of DMA pieces");
    static_assert(LDS_BYTES == patch_lds_bytes<NFRAG, KC, MI>(), "the entry points allocate what the tile function uses");

    TRTX_MARK(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, w_bytes, 0x00020000);

    // ---- the patch: every plane's pieces, dealt to the waves; padding ring, pitch padding and ragged channels are range-checked away (zero fill)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        for (int piece = wave; piece < PIECES; piece += 4) {
            const px::DmaLane d = px::dma_lane(piece, lane);
            const int hi = T.y0 - 1 + d.py, wi = T.x0 - 1 + d.px;
            const bool ok = d.px < px::kPW && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && kc * 32 + d.clog * 8 < p.Cin;
            const unsigned voff = ok ? (unsigned)((((T.n * p.H + hi) * p.W + wi) * p.ld_in + kc * 32 + d.clog * 8) * 2) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(smem + kc * PLANE + piece * 1024), 16, voff, 0, 0, 0);
        }
    }
    // ---- weight tiles: conv_igemm's B layout for 32-wide steps; k-step e = tap * KC + kc starts at k = 32 e of the packed row
    unsigned b_off[B_PASSES];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
        const px::WLane w = px::w_lane(j, wave, lane);
        b_off[j] = w.row < BN ? (unsigned)(((T.n0 + w.row) * p.Kpad + w.clog * 8) * 2) : kOOB;   // (kOOB + any k offset < 2^30 stays out of range)
    }
    int issued = 0;
    auto issue_w = [&](int stage) {
        char* sb = smem + PATCH_BYTES + stage * STAGE_B;
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) {
            const unsigned voff = issued < NK ? b_off[j] : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sb + (4 * j + wave) * 1024), 16, voff, 0, 0, 0);
            b_off[j] += 64;
        }
        ++issued;
    };

    floatx4 acc[MI][NFRAG];
    intx4 acci[MI][NFRAG];  // unused (the shared epilogue's int8 leg)
    floatx4 part[F32 ? MI : 1][F32 ? NFRAG : 1];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            acci[i][j] = intx4{0, 0, 0, 0};
            if constexpr (F32) part[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    if constexpr (F32) {
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) {
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(p.bias + T.n0 + j * 16 + (lane >> 4) * 4);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = b4;
            }
        }
    }
    int a_off[MI][3];   // fragment i of this wave = tile row wave * MI + i; tap column q; filter row r adds kRowStepBytes
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) a_off[i][q] = px::frag_offset(wave * MI + i, lane & 15, 0, q, lane >> 4);
    int fb_off[NFRAG];
#pragma unroll
    for (int j = 0; j < NFRAG; ++j) fb_off[j] = px::w_frag_offset(j, lane);

    issue_w(0);
    issue_w(1);
    // One k-step: the patch (first step) and weight tile e have landed once only tile e + 1's loads are in flight; every wave's fragment reads of
    // step e - 1 have COMPLETED (lgkmcnt(0): the k-step comment of the main kernel) before the barrier that frees their stage for tile e + 2.
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int e = (r * 3 + q) * KC + kc;
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B_PASSES) : "memory");
                __builtin_amdgcn_s_barrier();
                issue_w((e + 2) % NST);
                const char* pa = smem + kc * PLANE + r * px::kRowStepBytes;
                const char* pb = smem + PATCH_BYTES + (e % NST) * STAGE_B;
                if constexpr (F32) {
                    floatx4 af[MI], bf[NFRAG];
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const floatx4*>(pa + a_off[i][q]);
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) bf[j] = *reinterpret_cast<const floatx4*>(pb + fb_off[j]);
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            acc[i][j] += part[i][j];
                            part[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][0], af[i][0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        }
#pragma unroll
                    for (int s4 = 1; s4 < 4; ++s4)
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                            for (int i = 0; i < MI; ++i) part[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s4], af[i][s4], part[i][j], 0, 0, 0);
                } else {
                half8 af[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const half8*>(pa + a_off[i][q]);
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) {
                    const half8 bf = *reinterpret_cast<const half8*>(pb + fb_off[j]);
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, af[i], acc[i][j], 0, 0, 0);
                }
                }
            }
    // the two run-out weight tiles were range-checked away (no memory access) but their LDS writes must retire before the epilogue reuses the space
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // row t of the tile (t = 16 * tile row + column) -> output pixel, or -1 beyond the image
    auto pixel_of = [&](int t) {
        const int y = T.y0 + (t >> 4), x = T.x0 + (t & 15);
        return (y < p.H && x < p.W) ? (T.n * p.H + y) * p.W + x : -1;
    };
    if constexpr (F32) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) acc[i][j] += part[i][j];   // the last step's partial
        conv_epilogue_f32<NFRAG, MI>(p, acc, lane, T.n0, pixel_of, wave * MI * 16);
    } else {
        conv_epilogue<NFRAG, MI, false, LDS_BYTES>(p, acc, acci, smem, wave, lane, T.n0, pixel_of);
    }
    TRTX_MARK(3);
}

template <int NFRAG, int KC, int MI, bool F32 = false>
__global__ __launch_bounds__(256) void conv_patch_f16_kernel(const ConvArgs p, unsigned in_bytes, unsigned w_bytes, int tiles_n, int tiles_x, int tiles_y,
                                                             int total_tiles, int xcd_chunk) {
    __shared__ __attribute__((aligned(16))) char smem[patch_lds_bytes<NFRAG, KC, MI>()];
    int tile = blockIdx.x;
    if (xcd_chunk) {
        tile = (tile & 7) * xcd_chunk + (tile >> 3);
        if (tile >= total_tiles) return;
    }
    conv_patch_tile<NFRAG, KC, MI, F32>(p, in_bytes, w_bytes, patchidx::tile_of(tile, tiles_n, tiles_x, tiles_y, 4 * MI, 16 * NFRAG), smem);
}

}  // namespace
}  // namespace trtx
