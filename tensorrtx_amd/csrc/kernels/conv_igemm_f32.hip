// The convolutions of an fp32 engine on the matrix cores: NHWC fp32 activations, fp32 weights packed [Cout_pad][Kpad], v_mfma_f32_16x16x4_f32
// (fp32 operands, fp32 accumulate: bit-for-bit a k-ordered fmaf chain, 64 FLOP/clk/SIMD = 157 TFLOP/s on the chip - MI355X_MICROARCH.md).
//
// Why this exists.  The reference builds every model in one of three precisions (yolov8/include/config.h:1-3 USE_FP16 / USE_FP32 / USE_INT8,
// yolov8/src/model.cpp:314-324; retinaface/retina_r50.cpp:12), and BASELINE's tolerance - boxes within 1e-3 IoU, logits within 1e-4 of the
// reference's fp32 outputs - is a property of fp32 arithmetic: no engine that stores weights or activations in fp16 reaches it (DESIGN 2,
// profiles/r04_fp16_budget.txt).  Until round 5 the build without kFP16 ran conv_direct_kernel (nhwc_ops.hip): one thread per output, a scalar
// fmaf loop, no MFMA, no LDS.  This file puts that build on the implicit-GEMM skeleton of igemm_tile.h.
//
// How.  An fp32 k-step is 16 channels = a 64-byte LDS row - byte for byte the geometry of the fp16 kernel's 32-half step.  The launcher hands the
// kernel its input-side geometry in 2-byte units (Cin, ld_in, CinK, K, Kpad, up_C, up_ld doubled - what the int8 path does with pairs of int8
// channels), so the whole operand path is the fp16 one: range-checked LDS-DMA gather, XOR-swizzled rows, tap masks, three stages, one barrier
// per step, plain-GEMM addressing for 1x1 layers, the folded upsample, two filter taps per step for Cin <= 8.  What differs is compute() - one
// 16-byte fragment read feeds four MFMAs of 32 cycles each - and the epilogue, which needs no LDS: a lane's accumulator fragment is four
// consecutive channels of one pixel, a 16-byte fp32 store (conv_epilogue_f32).
// The kernel is MFMA-bound by construction: a 128 x 64 tile issues 32 MFMAs = 1024 cycles per k-step and wave against ~150 other instructions,
// 12 KB of LDS fill and 6 fragment reads; the 20 - 40 % of the fp16 kernels' time that is LDS fill, address arithmetic and barriers sits in the
// shadow of the matrix pipe here.  Every tile shape sums K in the same order: all tactics return the same bits.
#include "../options.h"
#include "igemm_tile.h"
#include "patch_tile.h"

namespace trtx {
namespace {

// kernel-side view of an fp32 launch: the input side in 2-byte units (see the header)
ConvArgs kernel_units(const ConvArgs& a) {
    ConvArgs k = a;
    k.Cin = 2 * a.Cin; k.ld_in = 2 * a.ld_in; k.CinK = 2 * a.CinK; k.K = 2 * a.K; k.Kpad = 2 * a.Kpad;
    k.up_C = 2 * a.up_C; k.up_ld = 2 * a.up_ld;
    k.bk = 2 * a.bk;
    return k;
}

bool plain_gemm_f32(const ConvArgs& a) {
    return a.kh == 1 && a.kw == 1 && a.stride_h == 1 && a.stride_w == 1 && a.pad_h == 0 && a.pad_w == 0 && a.Cin == a.CinK && a.Kpad == a.K && a.up_C == 0;
}

// operands through registers (global -> VGPR -> ds_write, two LDS stages) instead of LDS-DMA (three stages): ConvArgs::t_ws == 5, same bits.  Instantiated for
// the 64- and 128-row tiles of the general and the plain-GEMM walk
bool rs_exists(const ConvArgs& a, int bn, int bm) { return a.CinK != 8 && a.up_C == 0 && bm <= 128 && bn >= 32; }

// 32-channel k-steps (ConvArgs::bk == 32: 128-byte LDS rows = whole cache lines per row, two LDS stages): layers whose channels per tap are a multiple of 32;
// the same packed weights and the same bits as the 16-channel step.  Instantiated for the 64- and 128-row tiles, 32..128 columns wide, of the general and the
// plain-GEMM walk and of the folded upsample
bool wide_exists(const ConvArgs& a, int bn, int bm) { return a.CinK % 32 == 0 && a.Kpad % 32 == 0 && bm <= 128 && bn >= 32 && (a.up_C % 32) == 0; }

// roles (ConvArgs::t_ws == 6; igemm_tile.h ROLES): four fetching + four multiplying waves per workgroup, 16-channel steps through LDS-DMA, same bits.
// Instantiated for the 64- and 128-row tiles, 32..128 columns wide, of the general walk, the plain-GEMM walk and the folded upsample
bool roles_exist(const ConvArgs& a, int bn, int bm) { return a.CinK != 8 && bm <= 128 && bn >= 32; }

template <int NFRAG, int MI>
void launch_f32(const ConvArgs& a, const ConvArgs& k, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    const int BN = 16 * NFRAG, BMT = 64 * MI;
    const int tiles_m = (a.M + BMT - 1) / BMT, tiles_n = a.Cout_pad / BN;
    const int total = tiles_m * tiles_n, chunk = (total + 7) / 8;
    const dim3 grid(chunk * 8), block(256);
    if constexpr (MI <= 2 && NFRAG >= 2) {
        if (a.bk == 32) {
            if (a.up_C > 0)
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 64, 1, false, MI, 1, 0, false, 4, false, true, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
            else if (plain_gemm_f32(a))
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 64, 1, false, MI, 1, 0, false, 4, false, false, true, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
            else
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 64, 1, false, MI, 1, 0, false, 4, false, false, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
            return;
        }
    }
#ifdef TRTX_CONV_ABLATE   // timing experiments only (tools/f32_ablation.sh): TRTX_CONV_DBG as in the fp16 kernel, TRTX_F32_NST = 4 / 6 LDS stages (128 x 64 tile)
    const int dbg = options().conv_dbg, nst = options().f32_stages;
    if constexpr (MI == 2 && NFRAG == 4) {
        if (a.CinK != 8 && !a.up_C && !plain_gemm_f32(a) && a.t_ws != 5) {
            if (nst == 4) TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 4, false, 4, false, false, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, dbg);
            else if (nst == 6) TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 6, false, 4, false, false, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, dbg);
            else TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, false, false, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, dbg);
            return;
        }
    }
#endif
    if constexpr (MI <= 2 && NFRAG >= 2) {
        if (a.t_ws == 6 && roles_exist(a, BN, BMT)) {
            const dim3 block2(512);
            if (a.up_C > 0)
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, false, true, false, true, true>), grid, block2, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
            else if (plain_gemm_f32(a))
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, false, false, true, true, true>), grid, block2, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
            else
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, false, false, false, true, true>), grid, block2, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
            return;
        }
        if (a.t_ws == 5 && rs_exists(a, BN, BMT)) {
            if (plain_gemm_f32(a))
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, true, false, true, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
            else
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, true, false, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
            return;
        }
    }
    if (a.CinK == 8) {   // two filter taps per k-step (the 3-channel stems, padded to 4)
        if constexpr (MI == 2 && NFRAG <= 4)
            TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 2, false, MI, 1, 0, false, 4, false, false, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
    } else if (a.up_C > 0) {
        if constexpr (MI <= 2)
            TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, false, true, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
    } else if (plain_gemm_f32(a)) {
        TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, false, false, true, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
    } else {
        TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, 32, 1, false, MI, 1, 0, false, 4, false, false, false, true>), grid, block, 0, s, k, in_bytes, w_bytes, tiles_n, total, chunk, 0);
    }
}

// ---- the resident-patch kernel with fp32 operands (ConvArgs::t_ws == 3; patch_tile.h): 3x3 stride 1 pad 1, the input of a TH x 16 output block lies in LDS
// once as Cin / 16 planes, only the weight tile streams per k-step.  Planes: 16 / 32 / 48 / 64 / 80 / 128 input channels (1 .. 8 planes of an 8-row patch:
// 15 KB each; 16-row patches up to two planes), column tiles 16 .. 80 wide.  Same K order, same two-level sum: the implicit-GEMM kernel's bits.
bool patch_possible_f32(const ConvArgs& a, int bn) {
    const int kc = a.CinK / 16;
    return !a.up_C && a.kh == 3 && a.kw == 3 && a.stride_h == 1 && a.stride_w == 1 && a.pad_h == 1 && a.pad_w == 1 && a.CinK % 16 == 0 &&
           (kc <= 5 || kc == 8) && a.Kpad == 9 * a.CinK && !a.scalar_out && a.Ho == a.H && a.Wo == a.W &&
           (bn == 16 || bn == 32 || bn == 64 || bn == 80) && a.Cout_pad % bn == 0;   // (128 columns: 2 x 64 accumulator registers + 40 of fragments - no)
}
int patch_rows_f32(const ConvArgs& a) { return (a.CinK <= 32 && a.H >= 16) ? 16 : 8; }

template <int NFRAG, int KC>
void launch_patch_f32_mi(const ConvArgs& a, const ConvArgs& k, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    const int th = patch_rows_f32(a);
    const int tiles_n = a.Cout_pad / (16 * NFRAG), tiles_x = (a.W + 15) / 16, tiles_y = (a.H + th - 1) / th;
    const int total = a.N * tiles_y * tiles_x * tiles_n, chunk = (total + 7) / 8;
    if constexpr (KC <= 2) {
        if (th == 16) {
            TRTX_LAUNCH((conv_patch_f16_kernel<NFRAG, KC, 4, true>), dim3(chunk * 8), dim3(256), 0, s, k, in_bytes, w_bytes, tiles_n, tiles_x, tiles_y, total, chunk);
            return;
        }
    }
    TRTX_LAUNCH((conv_patch_f16_kernel<NFRAG, KC, 2, true>), dim3(chunk * 8), dim3(256), 0, s, k, in_bytes, w_bytes, tiles_n, tiles_x, tiles_y, total, chunk);
}
template <int NFRAG>
int32_t launch_patch_f32_kc(const ConvArgs& a, const ConvArgs& k, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    switch (a.CinK / 16) {
        case 1: launch_patch_f32_mi<NFRAG, 1>(a, k, in_bytes, w_bytes, s); break;
        case 2: launch_patch_f32_mi<NFRAG, 2>(a, k, in_bytes, w_bytes, s); break;
        case 3: launch_patch_f32_mi<NFRAG, 3>(a, k, in_bytes, w_bytes, s); break;
        case 4: launch_patch_f32_mi<NFRAG, 4>(a, k, in_bytes, w_bytes, s); break;
        case 5: launch_patch_f32_mi<NFRAG, 5>(a, k, in_bytes, w_bytes, s); break;
        case 8: launch_patch_f32_mi<NFRAG, 8>(a, k, in_bytes, w_bytes, s); break;
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return TRTX_OK;
}
int32_t launch_patch_f32(const ConvArgs& a, const ConvArgs& k, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    switch (a.bn) {
        case 16: return launch_patch_f32_kc<1>(a, k, in_bytes, w_bytes, s);
        case 32: return launch_patch_f32_kc<2>(a, k, in_bytes, w_bytes, s);
        case 64: return launch_patch_f32_kc<4>(a, k, in_bytes, w_bytes, s);
        case 80: return launch_patch_f32_kc<5>(a, k, in_bytes, w_bytes, s);
        default: return TRTX_ERR_UNSUPPORTED;
    }
}

// which (column-tile width, rows per tile) pairs are instantiated
bool tile_exists(const ConvArgs& a, int bn, int bm) {
    if (bn != 16 && bn != 32 && bn != 64 && bn != 80 && bn != 128) return false;
    if (bm != 64 && bm != 128 && bm != 256) return false;
    if (bm == 256 && bn == 128) return false;               // 128 accumulator registers + 48 of fragments: no
    if (a.CinK == 8) return bm == 128 && bn <= 64;
    if (a.up_C > 0) return bm <= 128;
    return true;
}

template <int MI>
int32_t launch_f32_bn(const ConvArgs& a, const ConvArgs& k, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    switch (a.bn) {
        case 16: launch_f32<1, MI>(a, k, in_bytes, w_bytes, s); break;
        case 32: launch_f32<2, MI>(a, k, in_bytes, w_bytes, s); break;
        case 64: launch_f32<4, MI>(a, k, in_bytes, w_bytes, s); break;
        case 80: launch_f32<5, MI>(a, k, in_bytes, w_bytes, s); break;
        case 128:
            if constexpr (MI <= 2) { launch_f32<8, MI>(a, k, in_bytes, w_bytes, s); break; }
            return TRTX_ERR_UNSUPPORTED;
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return TRTX_OK;
}

// The untuned tile of a layer (measured, profiles/r05_f32_shape_ab.txt: every shape lands within a few per cent of every other - the kernel's k-loop sustains
// ~0.6 of the matrix pipe whatever the tile - so the rule only has to avoid the bad corners): the widest column tile that divides the padded Cout (a narrower one
// re-reads the A tile Cout_pad / bn times; 64 instead of 128 when 128 would leave fewer than four tiles per CU), 64-row tiles (six workgroups per CU instead of
// four, half the tail) unless the map is so large that even 128-row tiles give sixteen tiles per CU.
void default_tile(const ConvArgs& a, int* bn_out, int* bm_out) {
    static const int bns[5] = {128, 80, 64, 32, 16};
    *bn_out = *bm_out = 0;
    for (int bn : bns) {
        if (a.Cout_pad % bn) continue;
        int bm = 64;
        if (!tile_exists(a, bn, bm)) bm = 128;
        if (!tile_exists(a, bn, bm)) continue;
        const long tiles64 = ((long)a.M + 63) / 64 * (a.Cout_pad / bn);
        if (bn == 128 && tiles64 < 4 * 256 && tile_exists(a, 64, bm)) continue;   // take the 64-wide tile
        if (tiles64 >= 32 * 256 && tile_exists(a, bn, 128)) bm = 128;
        *bn_out = bn;
        *bm_out = bm;
        return;
    }
}

}  // namespace

int conv_igemm_f32_pick_cink(int cin) { return cin <= 8 ? 8 : (cin + 15) / 16 * 16; }

bool conv_igemm_f32_supported(const ConvArgs& a) {
    if (!a.f32 || a.in_i8 || a.out_i8 || a.res_i8 || (a.bk != 16 && a.bk != 32)) return false;
    if (a.bk == 32 && !(a.bn && a.bm)) return false;   // the wide step is a named tactic (conv_tactics_f32), never the launcher's own choice
    if (a.groups != 1 || a.dil_h != 1 || a.dil_w != 1) return false;
    if (a.Cin % 4 || a.ld_in % 4) return false;                      // 16-byte channel chunks
    if (a.CinK != conv_igemm_f32_pick_cink(a.Cin) || a.K != a.kh * a.kw * a.CinK || a.Kpad != (a.K + 15) / 16 * 16) return false;
    if (a.CinK == 8 ? (a.kh > 30 || a.kw > 30) : (a.kh * a.kw > kMaxTaps)) return false;   // two taps per step: row and column masks are separate words
    if (a.Cout_pad % 16 || a.Cout_pad < a.Cout) return false;
    const bool out_vec = a.ld_out % 4 == 0 && a.Cout % 4 == 0 && (!a.residual || a.ld_res % 4 == 0);
    if (!out_vec && !a.scalar_out) return false;
    if (a.up_C != 0 && (a.up_C < 0 || a.kh != 1 || a.kw != 1 || a.stride_h != 1 || a.stride_w != 1 || a.pad_h || a.pad_w || a.up_C % 16 || a.up_C >= a.Cin ||
                        a.H != 2 * a.up_H || a.W != 2 * a.up_W || a.up_ld % 4 || a.CinK == 8))
        return false;
    if ((double)a.H * a.W * a.ld_in * 4.0 >= 2.0e9 || (double)a.Cout_pad * a.Kpad * 4.0 >= 2.0e9) return false;
    if (a.bn || a.bm) {   // a tactic was named
        const int bm = a.bm ? a.bm : 128;
        if (a.t_ws == 7) return a.bn && conv_res_possible(a);    // the resident-operand kernels (conv_res.hip) have their own shape tables
        if (a.t_ws == 8) return a.bn && conv_res1_possible(a);
        if (!a.bn || a.Cout_pad % a.bn || !tile_exists(a, a.bn, bm)) return false;
        if (a.t_ws == 5 && (!rs_exists(a, a.bn, bm) || a.bk != 16)) return false;
        if (a.t_ws == 6 && (!roles_exist(a, a.bn, bm) || a.bk != 16)) return false;
        if (a.t_ws == 3 && (!patch_possible_f32(a, a.bn) || a.bk != 16)) return false;
        if (a.bk == 32 && !wide_exists(a, a.bn, bm)) return false;
    } else {   // the launcher's own choice must exist (a folded upsample into 16 output channels has no instantiation)
        int bn = 0, bm = 0;
        default_tile(a, &bn, &bm);
        if (!bn) return false;
    }
    return true;
}

int conv_tactics_f32(const ConvArgs& a0, ConvTactic* out, int max_out) {
    ConvArgs a = a0;
    a.bn = 0; a.bm = 0;
    if (!conv_igemm_f32_supported(a)) return 0;
    int n = 0;
    auto push = [&](int bn, int bm, int ws = 1, int bk = 16) {
        for (int i = 0; i < n; ++i)
            if (out[i].bn == bn && out[i].bm == bm && out[i].ws == ws && out[i].bk == bk) return;
        if (n < max_out) out[n++] = ConvTactic{bn, bk, bm, 1, ws, 0};
    };
    int bn0, bm0;
    default_tile(a, &bn0, &bm0);
    push(bn0, bm0);
    static const int bns[5] = {128, 80, 64, 32, 16};
    static const int bms[3] = {128, 64, 256};
    for (int bn : bns)   // the resident-patch kernel at the widest column tile
        if (patch_possible_f32(a, bn)) {
            push(bn, 128, 3);
            break;
        }
    // the resident-operand 3x3 kernel (conv_res.hip, round 6; ws == 7): weights of a column tile resident in LDS, role-rotating wave groups - the same bits
    if (options().res & 1)
        for (int bn : bns) {
            ConvArgs t = a;
            t.bn = bn; t.bm = 128; t.t_ws = 7;
            if (a.Cout_pad % bn == 0 && conv_res_possible(t)) push(bn, 128, 7);
        }
    if (options().res & 2)   // ... and its 1x1 form (ws == 8)
        for (int bn : bns) {
            ConvArgs t = a;
            t.bn = bn; t.bm = 128; t.t_ws = 8;
            if (a.Cout_pad % bn == 0 && conv_res1_possible(t)) push(bn, 128, 8);
        }
    for (int bn : bns) {
        if (a.Cout_pad % bn) continue;
        if (bn <= 32 && a.Cout_pad > 2 * bn) continue;
        for (int bm : bms)
            if (tile_exists(a, bn, bm)) {
                push(bn, bm);
                if (wide_exists(a, bn, bm)) push(bn, bm, 1, 32);
                if (rs_exists(a, bn, bm)) push(bn, bm, 5);
                if (options().roles && roles_exist(a, bn, bm)) push(bn, bm, 6);
            }
    }
    return n;
}

int32_t conv_igemm_f32(const ConvArgs& a0, hipStream_t s) {
    if (!conv_igemm_f32_supported(a0)) return TRTX_ERR_UNSUPPORTED;
    if (a0.t_ws == 7 && a0.bn) return conv_res_f16(&a0, 1, s);   // (whole batch in one launch: its slice stays below 2 GB by conv_res_possible)
    if (a0.t_ws == 8 && a0.bn) return conv_res1_f16(a0, s);
    const size_t img_in = (size_t)a0.H * a0.W * a0.ld_in * 4;
    const int per = (int)std::max<size_t>(1, (size_t)2000000000 / img_in);
    const unsigned w_bytes = (unsigned)((size_t)a0.Cout_pad * a0.Kpad * 4);
    for (int n0 = 0; n0 < a0.N; n0 += per) {
        ConvArgs a = a0;
        a.N = std::min(per, a0.N - n0);
        a.M = a.N * a.Ho * a.Wo;
        a.in = static_cast<const char*>(a0.in) + (size_t)n0 * img_in;
        a.out = static_cast<char*>(a0.out) + (size_t)n0 * a.Ho * a.Wo * a.ld_out * 4;
        if (a0.residual) a.residual = static_cast<const char*>(a0.residual) + (size_t)n0 * a.Ho * a.Wo * a.ld_res * 4;
        if (a0.up_C) a.up_in = static_cast<const char*>(a0.up_in) + (size_t)n0 * a.up_H * a.up_W * a.up_ld * 4;
        if (!a.bn) default_tile(a, &a.bn, &a.bm);
        if (!a.bm) a.bm = 128;
        const ConvArgs k = kernel_units(a);
        const unsigned in_bytes = (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ld_in + a.Cin) * 4);
        int32_t st;
        if (a.t_ws == 3) st = launch_patch_f32(a, k, in_bytes, w_bytes, s);
        else if (a.bm == 64) st = launch_f32_bn<1>(a, k, in_bytes, w_bytes, s);
        else if (a.bm == 256) st = launch_f32_bn<4>(a, k, in_bytes, w_bytes, s);
        else st = launch_f32_bn<2>(a, k, in_bytes, w_bytes, s);
        if (st != TRTX_OK) return st;
    }
    return check_launch("conv_igemm_f32");
}

// fp32 [Cout_pad][Kpad], k = (r*kw + q)*cink + c, zero padded; the folded BatchNorm scale multiplied in (fp32)
void conv_pack_weights_igemm_f32(const float* w, int cout, int cin, int kh, int kw, int cink, int kpad, int cout_pad, const float* ch_scale, float* packed) {
    memset(packed, 0, sizeof(float) * (size_t)cout_pad * kpad);
    for (int co = 0; co < cout; ++co) {
        const float sc = ch_scale ? ch_scale[co] : 1.0f;
        for (int c = 0; c < cin; ++c)
            for (int r = 0; r < kh; ++r)
                for (int q = 0; q < kw; ++q)
                    packed[(size_t)co * kpad + (size_t)(r * kw + q) * cink + c] = w[(((size_t)co * cin + c) * kh + r) * kw + q] * sc;
    }
}

}  // namespace trtx
