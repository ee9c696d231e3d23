// Internal kernel-launcher interface (C++), used by the runtime executor and by the thin
// extern "C" test/bench entry points in ops_capi.cpp.  Everything here runs on a HIP stream;
// nothing falls back to the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace trtx {

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_SILU = 3, ACT_LEAKY = 4, ACT_TANH = 5,
                 ACT_MISH = 6 /* x * tanh(softplus(x)), the reference's Mish_TRT plugin (yolov4/mish.cu:113-135) */ };
enum DType : int { DT_F32 = 0, DT_F16 = 1, DT_I8 = 2 };
inline size_t dtype_size(int dt) { return dt == DT_F16 ? 2 : (dt == DT_I8 ? 1 : 4); }
enum EwOp : int { EW_SUM = 0, EW_PROD = 1, EW_MAX = 2, EW_MIN = 3, EW_SUB = 4, EW_DIV = 5, EW_POW = 6 };
enum PoolOp : int { POOL_MAX = 0, POOL_AVG = 1 };

// One fused convolution launch.  Activations are NHWC; `in`/`out`/`residual` already point at the
// first channel of the slice they address and ld_* is the channel stride of the underlying buffer.
struct ConvArgs {
    const void* in;
    const void* wgt;    // igemm: fp16 [Cout_pad][Kpad], k = (r*kw+q)*CinK + c.  direct: fp32 [Cout][kh*kw*Cin/groups]
    const float* bias;  // [Cout_pad] (folded BN shift / conv bias) or nullptr
    void* out;
    const void* residual;  // same geometry as out, or nullptr
    int N, H, W, Cin, ld_in;
    int Ho, Wo, Cout, Cout_pad, ld_out, ld_res;
    int kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups;
    int K, Kpad, M;
    int act1, act2;
    float alpha1, alpha2;
    int bn;  // igemm column-tile width (16/32/64/80/128)
    int scalar_out;  // igemm: element-wise epilogue stores (Cout, channel stride or offset not a multiple of 8)
    int CinK;        // igemm: per-tap stride of the packed K axis: Cin rounded up to bk, or 16 (two taps per step);
                     // Kpad = kh*kw*CinK rounded up to bk
    int bk;          // igemm: k-step width in halfs (32, or 64 when CinK % 64 == 0)
    // INT8 (kINT8 engines, conv_igemm only).  in_i8: activations and weights are int8, accumulated by v_mfma_i32_16x16x64_i8;
    // the input-side fields Cin, ld_in, CinK, K, Kpad are then counted in PAIRS of int8 channels (= 2-byte units), so that
    // every byte offset the kernel derives is the one the fp16 kernel would derive; Cout / ld_out / ld_res stay in channels.
    // out = act(acc * cscale[c] + bias[c]) with cscale[c] = input_scale * weight_scale[c]; out_i8: quantised with out_inv_scale
    // (round to nearest even, clamp +-127); res_i8: the residual is int8 with scale res_scale.
    int in_i8, out_i8, res_i8;
    const float* cscale;   // [Cout_pad]
    float out_inv_scale, res_scale;
    // Tactic (see ConvTactic): 0 everywhere = the untuned dispatch.
    int bm;     // igemm rows per tile: 0 / 128, 64 or 256
    int t_wsk;  // wave-split-K variant: 0 = by the static rule, 1 = never, 2 = wherever it exists
    int t_ws;   // weight-stationary kernel: 0 = where supported, 1 = never, 2 = asked for (still only where supported); 5 = (fp32 launches) operands through registers; 6 = (fp32 launches) fetching + multiplying wave roles; 3 = the resident-patch 3x3; 7 / 8 = the resident-operand 3x3 / 1x1 kernels (conv_res.hip)
                // kernel instead (conv_igemm.hip, builds with -DTRTX_EXPERIMENTAL_PATCH only)
    int t_rs;   // implicit-GEMM operands through registers (global -> VGPR -> ds_write) instead of LDS-DMA: 0 = no, 1 = yes (same bits)
    int t_r3;   // 3x3 stride-1 row-reuse kernel (conv_igemm_r3_f16_kernel, only where it exists): 0 = no, 1 = three LDS stages, 2 = two
    // Folded nearest 2x upsample (conv_igemm main kernel, 1x1 stride-1 convolutions only): input channels [0, up_C) are not read from
    // `in` but from `up_in`, an NHWC fp16 tensor of HALF the resolution ([N][up_H][up_W][up_ld], H = 2 up_H, W = 2 up_W), at pixel
    // (h >> 1, w >> 1); channels >= up_C come from `in` as usual.  This is Upsample -> Concat -> Conv1x1 (YOLOv8 head, model.cpp:130-160)
    // without the upsampled tensor ever existing.  up_C = 0: off.
    const void* up_in;
    int up_C, up_ld, up_H, up_W;
    // fp32 engines (builds without kFP16; conv_igemm_f32.hip): activations, residual and packed weights are fp32, the k-step is 16 channels
    // (bk == 16; CinK = Cin rounded up to 16, or 8 for Cin <= 8: two filter taps per step), Kpad = K rounded up to 16.  All fields in channels.
    int f32;
    // The layer's summation order is pinned to the main kernel's (lowering: members of a grouped launch, and the would-be members under TRTX_GROUP_CONVS=0):
    // conv_tactics() lists no wave-split-K / weight-stationary candidate for it - the tuner cannot hand it another K order.  Part of the tactic signature.
    int k_pinned;
};

// One launch configuration of an implicit-GEMM layer.  All tactics of a layer share its packed weights (Cout_pad, CinK, Kpad), so
// the executor may exchange them at run time: runtime/tune.cpp times them in place and keeps the fastest (TensorRT's builder does
// the same with its tactics; here it happens at deserializeCudaEngine because plans store the network, not kernels).
struct ConvTactic {
    int bn, bk, bm;  // column-tile width, k-step width, rows per tile
    int wsk, ws;     // values of ConvArgs::t_wsk / t_ws
    int r3;          // value of ConvArgs::t_r3
};

// Launch probe (IProfiler-style per-layer timing): while one is set for the calling thread, the first MFMA / stem convolution
// launch records the DISPATCH's own begin and end timestamps into its events (hipExtLaunchKernelGGL) - the kernel's duration as
// rocprofv3 sees it, without the hand-over between two stream events that bracketing a launch with hipEventRecord includes.
struct LaunchProbe {
    hipEvent_t start = nullptr, stop = nullptr;
    int launches = 0;
};
void conv_set_launch_probe(LaunchProbe* p);  // nullptr: off
LaunchProbe* conv_launch_probe();

// --- conv -------------------------------------------------------------------------------------------
// fused implicit-GEMM conv on MFMA (conv_igemm.hip): LDS-DMA operands, range-checked gather, 3-stage pipeline
int conv_igemm_pick_bn(int cout);   // column-tile width for a Cout
int conv_igemm_pick_bk(int cin, int taps);  // k-step width for a (padded-to-8) Cin and kh*kw filter taps
int conv_igemm_pick_cink(int cin, int bk);  // per-tap stride of the packed K axis
bool conv_igemm_supported(const ConvArgs& a);
// int8 weight packing: per-output-channel symmetric scales (max |w| / 127 after the BN scale is folded in), same [Cout_pad][Kpad]
// row layout as the fp16 packing with CinK = Cin rounded up to 64 channels; wscale_out[Cout_pad]
void conv_pack_weights_i8(const float* w_kcrs, int cout, int cin, int kh, int kw, int cink, const float* ch_scale, int cout_pad, int kpad,
                          int8_t* packed, float* wscale_out);
int32_t conv_igemm_f16(const ConvArgs& a, hipStream_t s);
// the tactics applicable to a layer (a.N / a.M at the batch it will run with); out[0] is the untuned default. Returns the count.
// work_efficient_only: leave out the configurations that buy latency with extra traffic (narrower column tiles re-read the
// activations, 64-row tiles re-read the weights) - what an engine whose contexts run side by side should choose from.
int conv_tactics(const ConvArgs& a, ConvTactic* out, int max_out, bool work_efficient_only = false);
void conv_apply_tactic(ConvArgs* a, const ConvTactic& t);
// Several independent fp16 implicit-GEMM convolutions in ONE launch (conv_igemm.hip conv_igemm_group_f16_kernel): the members share a
// kernel instantiation - 128-row tiles, one column-tile width (64 or 80), 32-wide k-steps, the same operand path (t_rs) and all or none of
// them plain 1x1 GEMMs - and each is computed exactly as its own launch would compute it (same tiles, same K order: the same bits).
constexpr int kMaxConvGroup = 4;
bool conv_igemm_group_supported(const ConvArgs* a, int n);
int32_t conv_igemm_group_f16(const ConvArgs* a, int n, hipStream_t s);
// Large plain-GEMM convolutions on the 256 x 256 x 64 role-alternating tile (conv_gemm256.hip): the tactic ConvArgs::bn == 256 of the
// implicit-GEMM family (same packed weights, same bits); conv_igemm_f16 dispatches to it
bool conv_gemm256_possible(const ConvArgs& a);     // may run (any batch up to the build batch)
bool conv_gemm256_worthwhile(const ConvArgs& a);   // is a candidate of the tactic timing at this batch
int32_t conv_gemm256_f16(const ConvArgs& a, hipStream_t s);
// fp32 engines: the same skeleton on v_mfma_f32_16x16x4_f32 (conv_igemm_f32.hip).  ConvArgs::f32 = 1, bn / bm = 0: the launcher's own tile choice
int conv_igemm_f32_pick_cink(int cin);
bool conv_igemm_f32_supported(const ConvArgs& a);
int32_t conv_igemm_f32(const ConvArgs& a, hipStream_t s);
int conv_tactics_f32(const ConvArgs& a, ConvTactic* out, int max_out);   // (bn, bm) pairs; out[0] = the untuned choice; every pair returns the same bits
void conv_pack_weights_igemm_f32(const float* w_kcrs, int cout, int cin, int kh, int kw, int cink, int kpad, int cout_pad, const float* ch_scale, float* packed);
// Resident-operand 3x3 stride-1 kernel (conv_res.hip, round 6; tactic ConvArgs::t_ws == 7): persistent workgroups keep the layer's whole weight slab in LDS, two
// role-alternating halves of 4 waves (k-loop | register epilogue + next patch fetch); same packed weights, same K order, same bits as the main kernel.
// a[0..n): 1..kMaxConvGroup independent layers of one instantiation in one launch (a workgroup is bound to one of them).
bool conv_res_possible(const ConvArgs& a);
bool conv_res_group_possible(const ConvArgs* a, int n);
int32_t conv_res_f16(const ConvArgs* a, int n, hipStream_t s);
// ... and its 1x1 stride-1 sibling (tactic t_ws == 8): the column tile's weights resident in LDS, 16 independent waves per workgroup, the A operand global -> VGPR
bool conv_res1_possible(const ConvArgs& a);
int32_t conv_res1_f16(const ConvArgs& a, hipStream_t s);
// weight-stationary persistent kernel for small-channel 3x3 (stride 1, pad 1) and 1x1 layers (conv_ws.hip): weights in
// registers, input patch staged once in LDS; same packed weights / ConvArgs as the implicit-GEMM kernel, which dispatches to it
bool conv_ws_supported(const ConvArgs& a);
int32_t conv_ws_f16(const ConvArgs& a, hipStream_t s);
int32_t poison_lds(unsigned* device_word, hipStream_t s);  // test support (test_support.hip): NaN patterns into every CU's LDS
// first layer: fp32 NCHW input (1..4 channels) -> NHWC fp16, weights fp32 [kh*kw*Cin (c,r,q)][Cout]
bool conv_stem_supported(const ConvArgs& a);
int32_t conv_stem_nchw_f32(const ConvArgs& a, hipStream_t s);
// ... and of an fp32 engine (conv_stem_f32.hip): the same input and weight layout, NHWC fp32 out, Cout a multiple of 16
bool conv_stem_f32_supported(const ConvArgs& a);
int32_t conv_stem_nchw_f32_out_f32(const ConvArgs& a, hipStream_t s);
// The stem fed by camera frames instead of the fp32 network input: image n of the batch is the letterbox (yolov8/src/preprocess.cu) of
// frames[n] - a device-resident uint8 HWC BGR image of w x h pixels with d2s = the inverse affine map trtx_letterbox_matrix computes
// for (w, h) -> (a.W, a.H) - sampled inside the kernel's patch fill.  a.in is ignored.  3-channel stems on the LDS / MFMA path only
// (TRTX_ERR_UNSUPPORTED otherwise: the caller then runs the letterbox kernel and the ordinary stem).
constexpr int kStemMaxFrames = 64;
struct StemFrame {
    const void* src;
    int w, h;
    float d2s[6];
};
int32_t conv_stem_frames_f32(const ConvArgs& a, const StemFrame* frames, hipStream_t s);
// generic direct convolution (any groups / dilation / channel count), T = activation dtype, fp32 weights
int32_t conv_direct(const ConvArgs& a, int dtype, hipStream_t s);
// generic transposed convolution, fp32 weights laid out [Cin][kh][kw][Cout/groups]
int32_t deconv_direct(const ConvArgs& a, int dtype, hipStream_t s);

// --- layout / dtype conversion ----------------------------------------------------------------------
// LINEAR fp32 [N][C][H][W] -> NHWC dtype, channels [C, Cpad) zero-filled
int32_t nchw_f32_to_nhwc(const float* in, void* out, int dtype, int N, int C, int H, int W, int Cpad, int ld_out,
                         hipStream_t s);
// NHWC dtype -> LINEAR fp32 [N][C][H][W]
int32_t nhwc_to_nchw_f32(const void* in, int dtype, float* out, int N, int C, int H, int W, int ld_in, hipStream_t s);

// --- NHWC element ops (dtype = DT_F16 / DT_F32), strided channel slices ---------------------------------
// SPPF: three chained k x k stride-1 'same' max-pools (fp16, C % 8 == 0, H*W*32 B <= 60 KB of LDS)
// (f32: fp32 NHWC tensors, 4-channel chunks - the fp32 engines' SPPF)
int32_t nhwc_maxpool_chain3_f16(const void* in, void* o1, void* o2, void* o3, int N, int H, int W, int C, int ld_in, int ld1,
                                int ld2, int ld3, int k, hipStream_t s, int f32 = 0);
// [N,H,W,(r,q,c)] -> [N,H*bh,W*bw,c], fp16, C % 8 == 0
int32_t nhwc_depth_to_space_f16(const void* in, void* out, int N, int H, int W, int C, int bh, int bw, int ld_in, int ld_out,
                                hipStream_t s);
int32_t nhwc_pool(const void* in, void* out, int dtype, int op, int N, int H, int W, int C, int ld_in, int Ho, int Wo,
                  int ld_out, int kh, int kw, int sh, int sw, int ph, int pw, int avg_exclusive, hipStream_t s);
int32_t nhwc_resize_nearest(const void* in, void* out, int dtype, int N, int H, int W, int C, int ld_in, int Ho,
                            int Wo, int ld_out, hipStream_t s);
int32_t nhwc_elementwise(const void* a, const void* b, void* out, int dtype, int op, long pixels, int C, int ld_a,
                         int ld_b, int ld_out, hipStream_t s);
int32_t nhwc_activation(const void* in, void* out, int dtype, int act, float alpha, long pixels, int C, int ld_in,
                        int ld_out, hipStream_t s);
int32_t nhwc_scale(const void* in, void* out, int dtype, const float* scale, const float* shift, long pixels, int C,
                   int ld_in, int ld_out, hipStream_t s);
int32_t nhwc_copy(const void* in, void* out, int dtype, long pixels, int C, int ld_in, int ld_out, hipStream_t s);
// mean over H*W: NHWC [N][H][W][C] -> NHWC [N][1][1][C]
int32_t nhwc_reduce_hw_avg(const void* in, void* out, int dtype, int N, int HW, int C, int ld_in, int ld_out,
                           hipStream_t s);

// --- INT8 support (quant_ops.hip): calibration statistics over NHWC fp16 tensors, int8 resize with requantisation
int32_t nhwc_absmax_f16(const void* x, long pixels, int C, int ld, unsigned* out_float_bits, hipStream_t s);  // atomicMax of the float bits
int32_t nhwc_hist_f16(const void* x, long pixels, int C, int ld, float range, unsigned long long* hist2048, hipStream_t s);
int32_t nhwc_resize_nearest_i8(const void* in, void* out, int N, int H, int W, int C, int ld_in, int Ho, int Wo, int ld_out, float ratio,
                               hipStream_t s);

// --- LINEAR fp32 ops (rank <= 6, row-major, batch outermost) -----------------------------------------------
struct StridedView {
    int rank;
    long shape[6];
    long stride_in[6];   // element strides into the source (0 = broadcast)
    long stride_in2[6];  // second operand (elementwise only)
};
// out (dense row-major `shape`) = in[gather by stride_in]   (permute / slice / broadcast copy)
int32_t lin_gather(const float* in, float* out, const StridedView& v, hipStream_t s);
// dense source -> strided destination (concat placement): out[stride_in-indexed] = in
int32_t lin_scatter(const float* in, float* out, const StridedView& v, hipStream_t s);
int32_t lin_elementwise(const float* a, const float* b, float* out, int op, const StridedView& v, hipStream_t s);
int32_t lin_activation(const float* in, float* out, int act, float alpha, long n, hipStream_t s);
int32_t lin_softmax(const float* in, float* out, long outer, long axis, long inner, hipStream_t s);
// C[b][m][n] = sum_k A(b,m,k) * B(b,k,n); ta/tb: operand stored transposed; batch stride 0 = broadcast
int32_t lin_matmul(const float* A, const float* B, float* C, int batch, int M, int N, int K, int ta, int tb,
                   long bsA, long bsB, hipStream_t s);
int32_t lin_reduce(const float* in, float* out, int op /*0 sum,1 avg,2 max*/, long outer, long axis, long inner,
                   hipStream_t s);
int32_t lin_scale(const float* in, float* out, const float* scale, const float* shift, const float* power, int mode,
                  long outer, long C, long inner, hipStream_t s);

}  // namespace trtx
