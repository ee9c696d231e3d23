// extern "C" entry points that expose single kernels (for parity tests, micro-benchmarks and
// reference-style plugin classes) plus the library-level queries.  See include/trtx_hip.h.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "options.h"
#include "kernels/kernels.h"
#include "runtime/pack.h"

using namespace trtx;

extern "C" const char* trtx_status_string(int32_t s) {
    switch (s) {
        case TRTX_OK: return "ok";
        case TRTX_ERR_INVALID: return "invalid argument";
        case TRTX_ERR_HIP: return "HIP error";
        case TRTX_ERR_WORKSPACE: return "workspace too small";
        case TRTX_ERR_UNSUPPORTED: return "unsupported configuration";
        case TRTX_ERR_NO_DEVICE: return "no HIP device";
        case TRTX_ERR_IO: return "I/O or parse error";
        case TRTX_ERR_STATE: return "invalid call order";
        default: return "unknown status";
    }
}

extern "C" int32_t trtx_abi_version(void) {
    return TRTX_ABI_VERSION;
}

extern "C" int32_t trtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int32_t trtx_conv_packed_dims(int cout, int cin_pad, int kh, int kw, int32_t* cout_pad, int32_t* kpad,
                                         int32_t* bn) {
    if (cout < 1 || cin_pad < 1 || kh < 1 || kw < 1) return TRTX_ERR_INVALID;
    const int b = conv_igemm_pick_bn(cout);
    if (bn) *bn = b;
    if (cout_pad) *cout_pad = (cout + b - 1) / b * b;
    if (kpad) {
        const int bk = conv_igemm_pick_bk(cin_pad, kh * kw);
        *kpad = (kh * kw * conv_igemm_pick_cink(cin_pad, bk) + bk - 1) / bk * bk;
    }
    return TRTX_OK;
}

extern "C" int32_t trtx_conv_pack_weights_f16(const float* w_kcrs, int cout, int cin, int kh, int kw, int cin_pad,
                                              const float* ch_scale, uint16_t* packed) {
    if (!w_kcrs || !packed || cin_pad < cin) return TRTX_ERR_INVALID;
    const int bk = conv_igemm_pick_bk(cin_pad, kh * kw);
    pack_conv_weights_f16(w_kcrs, cout, cin, kh, kw, conv_igemm_pick_cink(cin_pad, bk), bk, ch_scale, packed);
    return TRTX_OK;
}

// geometry of the launch trtx_op_conv2d_nhwc_f16 performs (pointers left null)
static ConvArgs op_conv_args(int N, int H, int W, int Cin, int ld_in, int Cout, int ld_out, int kh, int kw, int sh, int sw, int ph, int pw,
                             int act1, int has_res, int ld_res, int act2) {
    ConvArgs a{};
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.ld_in = ld_in;
    a.Ho = (H + 2 * ph - kh) / sh + 1;
    a.Wo = (W + 2 * pw - kw) / sw + 1;
    a.Cout = Cout;
    a.bn = conv_igemm_pick_bn(Cout);
    a.Cout_pad = (Cout + a.bn - 1) / a.bn * a.bn;
    a.ld_out = ld_out; a.ld_res = ld_res;
    a.kh = kh; a.kw = kw; a.stride_h = sh; a.stride_w = sw; a.pad_h = ph; a.pad_w = pw; a.dil_h = 1; a.dil_w = 1;
    a.groups = 1;
    a.bk = conv_igemm_pick_bk(Cin, kh * kw);
    a.CinK = conv_igemm_pick_cink(Cin, a.bk);
    a.K = kh * kw * a.CinK;
    a.Kpad = (a.K + a.bk - 1) / a.bk * a.bk;
    a.M = N * a.Ho * a.Wo;
    a.act1 = act1; a.act2 = act2; a.alpha1 = 0.1f; a.alpha2 = 0.1f;
    a.scalar_out = (Cout % 8 || ld_out % 8 || (has_res && ld_res % 8)) ? 1 : 0;
    return a;
}

static bool g_force_tactic = false;
static ConvTactic g_forced{};

extern "C" int32_t trtx_op_conv_force_tactic(const int32_t* t) {
    g_force_tactic = t != nullptr;
    if (t) g_forced = ConvTactic{t[0], t[1], t[2], t[3], t[4], t[5]};
    return TRTX_OK;
}

extern "C" int32_t trtx_op_conv2d_tactics(int N, int H, int W, int Cin, int ld_in, int Cout, int ld_out, int kh, int kw, int sh, int sw, int ph,
                                          int pw, int has_residual, int ld_res, int32_t* out6, int32_t max_out) {
    if (!out6 || max_out < 1 || N < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1) return 0;
    ConvArgs a = op_conv_args(N, H, W, Cin, ld_in, Cout, ld_out, kh, kw, sh, sw, ph, pw, 0, has_residual, ld_res, 0);
    a.residual = has_residual ? reinterpret_cast<const void*>(1) : nullptr;
    std::vector<ConvTactic> t(max_out);
    const int n = conv_tactics(a, t.data(), max_out);
    for (int i = 0; i < n; ++i) {
        out6[6 * i + 0] = t[i].bn; out6[6 * i + 1] = t[i].bk; out6[6 * i + 2] = t[i].bm; out6[6 * i + 3] = t[i].wsk; out6[6 * i + 4] = t[i].ws;
        out6[6 * i + 5] = t[i].r3;
    }
    return n;
}

extern "C" int32_t trtx_op_conv2d_nhwc_f16(const void* in, int N, int H, int W, int Cin, int ld_in, const void* wpacked,
                                           const float* bias, void* out, int Cout, int ld_out, int kh, int kw, int sh,
                                           int sw, int ph, int pw, int act1, const void* residual, int ld_res,
                                           int act2, trtx_stream_t stream) {
    ConvArgs a = op_conv_args(N, H, W, Cin, ld_in, Cout, ld_out, kh, kw, sh, sw, ph, pw, act1, residual != nullptr, ld_res, act2);
    a.in = in;
    a.wgt = wpacked;
    a.bias = bias;
    a.out = out;
    a.residual = residual;
    if (reinterpret_cast<uintptr_t>(out) & 15) a.scalar_out = 1;
    if (g_force_tactic) conv_apply_tactic(&a, g_forced);
    // TRTX_OP_REPS=n (timing tools only): n back-to-back launches per call, so that the device — not the Python caller — sets the pace
    const int reps = options().op_reps;
    int32_t st = TRTX_OK;
    for (int r = 0; r < reps && st == TRTX_OK; ++r) st = conv_igemm_f16(a, stream);
    return st;
}

// --- fp32 engines: the implicit-GEMM convolution on the fp32 MFMA (kernels/conv_igemm_f32.hip), test / tool entry points
static ConvArgs op_conv_args_f32(int N, int H, int W, int Cin, int ld_in, int Cout, int ld_out, int kh, int kw, int sh, int sw, int ph, int pw,
                                 int act1, int has_res, int ld_res, int act2) {
    ConvArgs a{};
    a.f32 = 1;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.ld_in = ld_in;
    a.Ho = (H + 2 * ph - kh) / sh + 1;
    a.Wo = (W + 2 * pw - kw) / sw + 1;
    a.Cout = Cout;
    a.Cout_pad = (Cout + 15) / 16 * 16;
    a.ld_out = ld_out; a.ld_res = ld_res;
    a.kh = kh; a.kw = kw; a.stride_h = sh; a.stride_w = sw; a.pad_h = ph; a.pad_w = pw; a.dil_h = 1; a.dil_w = 1;
    a.groups = 1;
    a.bk = 16;
    a.CinK = conv_igemm_f32_pick_cink(Cin);
    a.K = kh * kw * a.CinK;
    a.Kpad = (a.K + 15) / 16 * 16;
    a.M = N * a.Ho * a.Wo;
    a.act1 = act1; a.act2 = act2; a.alpha1 = 0.1f; a.alpha2 = 0.1f;
    a.scalar_out = (Cout % 4 || ld_out % 4 || (has_res && ld_res % 4)) ? 1 : 0;
    return a;
}

extern "C" int32_t trtx_conv_packed_dims_f32(int cout, int cin_pad, int kh, int kw, int32_t* cout_pad, int32_t* kpad, int32_t* cink) {
    if (cout < 1 || cin_pad < 1 || kh < 1 || kw < 1) return TRTX_ERR_INVALID;
    const int ck = conv_igemm_f32_pick_cink(cin_pad);
    if (cout_pad) *cout_pad = (cout + 15) / 16 * 16;
    if (kpad) *kpad = (kh * kw * ck + 15) / 16 * 16;
    if (cink) *cink = ck;
    return TRTX_OK;
}

extern "C" int32_t trtx_conv_pack_weights_f32(const float* w_kcrs, int cout, int cin, int kh, int kw, int cin_pad, const float* ch_scale, float* packed) {
    if (!w_kcrs || !packed || cin_pad < cin) return TRTX_ERR_INVALID;
    const int ck = conv_igemm_f32_pick_cink(cin_pad);
    conv_pack_weights_igemm_f32(w_kcrs, cout, cin, kh, kw, ck, (kh * kw * ck + 15) / 16 * 16, (cout + 15) / 16 * 16, ch_scale, packed);
    return TRTX_OK;
}

extern "C" int32_t trtx_op_conv2d_tactics_f32(int N, int H, int W, int Cin, int ld_in, int Cout, int ld_out, int kh, int kw, int sh, int sw, int ph,
                                              int pw, int has_residual, int ld_res, int32_t* out2, int32_t max_out) {   // (4 ints per entry)
    if (!out2 || max_out < 1 || N < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1) return 0;
    ConvArgs a = op_conv_args_f32(N, H, W, Cin, ld_in, Cout, ld_out, kh, kw, sh, sw, ph, pw, 0, has_residual, ld_res, 0);
    a.residual = has_residual ? reinterpret_cast<const void*>(1) : nullptr;
    std::vector<ConvTactic> t(max_out);
    const int n = conv_tactics_f32(a, t.data(), max_out);
    for (int i = 0; i < n; ++i) {
        out2[4 * i + 0] = t[i].bn;
        out2[4 * i + 1] = t[i].bm;
        out2[4 * i + 2] = t[i].ws;
        out2[4 * i + 3] = t[i].bk;
    }
    return n;
}

extern "C" int32_t trtx_op_conv2d_nhwc_f32(const void* in, int N, int H, int W, int Cin, int ld_in, const void* wpacked, const float* bias, void* out,
                                           int Cout, int ld_out, int kh, int kw, int sh, int sw, int ph, int pw, int act1, const void* residual,
                                           int ld_res, int act2, const int32_t* tile2, trtx_stream_t stream) {
    ConvArgs a = op_conv_args_f32(N, H, W, Cin, ld_in, Cout, ld_out, kh, kw, sh, sw, ph, pw, act1, residual != nullptr, ld_res, act2);
    a.in = in;
    a.wgt = wpacked;
    a.bias = bias;
    a.out = out;
    a.residual = residual;
    if ((reinterpret_cast<uintptr_t>(out) & 15) || (residual && (reinterpret_cast<uintptr_t>(residual) & 15))) a.scalar_out = 1;
    if (tile2) {
        a.bn = tile2[0];
        a.bm = tile2[1];
        a.t_ws = tile2[2];
        a.bk = tile2[3];
    }
    const int reps = options().op_reps;   // timing tools only
    int32_t st = TRTX_OK;
    for (int r = 0; r < reps && st == TRTX_OK; ++r) st = conv_igemm_f32(a, stream);
    return st;
}

// --- kINT8 conv, test / tool entry points.  Packed dims: cink = Cin rounded up to 64 channels, kpad = kh*kw*cink (bytes per row).
extern "C" int32_t trtx_conv_pack_weights_i8(const float* w_kcrs, int cout, int cin, int kh, int kw, const float* ch_scale, int8_t* packed,
                                             float* wscale_out, int32_t* cout_pad_out, int32_t* kpad_out) {
    if (!w_kcrs || cout < 1 || cin < 1) return TRTX_ERR_INVALID;
    const int bn = conv_igemm_pick_bn(cout);
    const int cout_pad = (cout + bn - 1) / bn * bn, cink = (cin + 63) / 64 * 64, kpad = kh * kw * cink;
    if (cout_pad_out) *cout_pad_out = cout_pad;
    if (kpad_out) *kpad_out = kpad;
    if (packed && wscale_out) conv_pack_weights_i8(w_kcrs, cout, cin, kh, kw, cink, ch_scale, cout_pad, kpad, packed, wscale_out);
    return TRTX_OK;
}

// in: int8 NHWC [N,H,W,ld_in]; out: int8 (out_inv_scale > 0) or fp16 NHWC; cscale[Cout_pad] = input scale * weight scale
extern "C" int32_t trtx_op_conv2d_nhwc_i8(const void* in, int N, int H, int W, int Cin, int ld_in, const void* wpacked, const float* cscale,
                                          const float* bias, void* out, int out_is_i8, float out_inv_scale, int Cout, int ld_out, int kh,
                                          int kw, int sh, int sw, int ph, int pw, int act1, const void* residual, int res_is_i8,
                                          float res_scale, int ld_res, int act2, trtx_stream_t stream) {
    if (Cin % 16 || ld_in % 16) return TRTX_ERR_INVALID;
    ConvArgs a{};
    a.in = in; a.wgt = wpacked; a.bias = bias; a.out = out; a.residual = residual;
    a.N = N; a.H = H; a.W = W;
    a.Ho = (H + 2 * ph - kh) / sh + 1;
    a.Wo = (W + 2 * pw - kw) / sw + 1;
    a.Cout = Cout;
    a.bn = conv_igemm_pick_bn(Cout);
    a.Cout_pad = (Cout + a.bn - 1) / a.bn * a.bn;
    a.ld_out = ld_out; a.ld_res = ld_res;
    a.kh = kh; a.kw = kw; a.stride_h = sh; a.stride_w = sw; a.pad_h = ph; a.pad_w = pw; a.dil_h = 1; a.dil_w = 1; a.groups = 1;
    a.bk = 32;
    const int cink = (Cin + 63) / 64 * 64;
    a.in_i8 = 1; a.out_i8 = out_is_i8; a.res_i8 = res_is_i8;
    a.cscale = cscale; a.out_inv_scale = out_inv_scale; a.res_scale = res_scale;
    // input-side geometry in 2-byte units (pairs of int8 channels)
    a.Cin = Cin / 2; a.ld_in = ld_in / 2; a.CinK = cink / 2;
    a.K = kh * kw * a.CinK;
    a.Kpad = a.K;
    a.M = N * a.Ho * a.Wo;
    a.act1 = act1; a.act2 = act2; a.alpha1 = 0.1f; a.alpha2 = 0.1f;
    a.scalar_out = 0;
    if (Cout % 8 || ld_out % 8 || (residual && ld_res % 8)) return TRTX_ERR_UNSUPPORTED;
    return conv_igemm_f16(a, stream);
}

extern "C" int32_t trtx_op_nchw_f32_to_nhwc_f16(const float* in, void* out, int N, int C, int H, int W, int Cpad,
                                                int ld_out, trtx_stream_t stream) {
    return nchw_f32_to_nhwc(in, out, DT_F16, N, C, H, W, Cpad, ld_out, stream);
}

extern "C" int32_t trtx_op_nhwc_f16_to_nchw_f32(const void* in, float* out, int N, int C, int H, int W, int ld_in,
                                                trtx_stream_t stream) {
    return nhwc_to_nchw_f32(in, DT_F16, out, N, C, H, W, ld_in, stream);
}

// test support: fill the LDS of every CU with fp16 NaN patterns (LDS survives kernel boundaries; tests/test_gpu_multi_context.py)
extern "C" int32_t trtx_op_poison_lds(void* device_word, trtx_stream_t stream) {
    if (!device_word) return TRTX_ERR_INVALID;
    return poison_lds(static_cast<unsigned*>(device_word), static_cast<hipStream_t>(stream));
}

