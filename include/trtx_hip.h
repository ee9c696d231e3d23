/*
 * trtx_hip.h — C ABI of libtrtx_hip.so, the MI355X (gfx950) runtime behind the tensorrtx
 * network-definition / plugin surface.
 *
 * Plain C: opaque handles, plain pointers and sizes, int32_t status returns.  No C++ or torch types.
 * Device pointers are ordinary hipMalloc'd addresses owned by the caller; `stream` is a hipStream_t.
 *
 * Section 1 (this part): the detection plugin operators — each entry point replaces the `enqueue`
 * body of one reference plugin (file:line cited per function) and is what a cgo/JNI/ctypes binding
 * or a reference-style IPluginV2::enqueue override would call.
 * Section 2: network-definition builder / engine / execution-context API (mirrors the nvinfer1
 * objects the reference host code uses; see include/NvInfer.h for the C++ shim on top of it).
 */
#ifndef TRTX_HIP_H_
#define TRTX_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without dragging the HIP headers into pure-C callers. */
#ifndef TRTX_NO_HIP_TYPES
struct ihipStream_t;
typedef struct ihipStream_t* trtx_stream_t;
#endif

/* ---- status codes ---------------------------------------------------------------------------- */
#define TRTX_ABI_VERSION 1
#define TRTX_OK 0
#define TRTX_ERR_INVALID 1     /* bad argument */
#define TRTX_ERR_HIP 2         /* a HIP call or kernel launch failed (message on stderr) */
#define TRTX_ERR_WORKSPACE 3   /* workspace too small */
#define TRTX_ERR_UNSUPPORTED 4 /* configuration outside what the kernels implement */
#define TRTX_ERR_NO_DEVICE 5   /* no gfx950 device visible: the runtime never falls back to the CPU */
#define TRTX_ERR_IO 6          /* file / parse error (.wts, plan) */
#define TRTX_ERR_STATE 7       /* call order violated (e.g. enqueue before build) */

const char* trtx_status_string(int32_t status);
/* ABI version of this header; bumped on any incompatible change. */
int32_t trtx_abi_version(void);
/* Number of visible HIP devices (0 when none; never an error). */
int32_t trtx_device_count(void);

/* ============================== Section 1: plugin operators ================================== */

/*
 * YOLOv8 anchor-free decode.  Replaces YoloLayerPlugin::enqueue -> forwardGpu -> CalDetection
 * (reference yolov8/plugin/yololayer.cu:167-172, 178-220, 282-316).
 *   inputs[l]  device, fp32 [batch][4+classes][ (net_h/strides[l]) * (net_w/strides[l]) ]  (CHW linear)
 *   output     device, fp32 [batch][1 + max_out*90]; output[b][0] = candidate count (clamped to
 *              max_out), then Detection records of 90 floats (yolov8/include/types.h:4-12) of which
 *              bbox[4] (xyxy), conf, class_id are written.  Candidates appear in (level, cell) order.
 *   `inputs` and `strides` are HOST arrays (as in IPluginV2::enqueue's `inputs`).
 */
size_t trtx_yolo_decode_workspace(int batch, int net_h, int net_w, const int* strides, int n_levels);
int32_t trtx_yolo_decode(const float* const* inputs, int n_levels, int batch, int classes, int net_h, int net_w,
                         const int* strides, int max_out, float* output, void* workspace, size_t workspace_bytes,
                         trtx_stream_t stream);

/* trtx_yolo_decode with the optional branches of CalDetection (yololayer.cu:222-279): inputs [batch][4 + classes (+32 seg)
 * (+3*n_kpt pose) (+1 obb)][cells]; seg copies the 32 mask coefficients into Detection::mask, pose decodes n_kpt keypoints
 * ((2*v + col|row) * stride, sigmoid confidence, -1 triple when below kpt_conf or outside the box), obb rotates the box
 * centre by (sigmoid(a) - 0.25) * pi and stores cx, cy, w, h + Detection::angle. */
int32_t trtx_yolo_decode_ex(const float* const* inputs, int n_levels, int batch, int classes, int net_h, int net_w,
                            const int* strides, int max_out, int n_kpt, float kpt_conf, int is_seg, int is_pose, int is_obb,
                            float* output, void* workspace, size_t workspace_bytes, trtx_stream_t stream);
/*
 * Fused DFL + YOLOv8 decode on the detect head's own NHWC fp16 output (what the engine uses instead of the
 * shuffle/slice/softmax/conv/concat chain of yolov8/src/block.cpp:239-257 + model.cpp:263-303 followed by
 * YoloLayerPlugin::enqueue).  heads[l]: device fp16 [batch][cells_l][ld[l]], channels [0,64) = 4x16 DFL
 * bins, [64, 64+classes) = class logits; dfl_weights: device fp32[16] (model.22.dfl.conv.weight).
 * Output format and candidate order are those of trtx_yolo_decode.  classes % 8 == 0.
 */
size_t trtx_yolo_head_decode_workspace(int batch, int net_h, int net_w, const int* strides, int n_levels);
int32_t trtx_yolo_head_decode_nhwc(const void* const* heads, const int* ld, int n_levels, int batch, int classes,
                                   int net_h, int net_w, const int* strides, const float* dfl_weights, int max_out,
                                   float* output, void* workspace, size_t workspace_bytes, trtx_stream_t stream);
/* The same on the NHWC fp32 head tensors of an fp32 engine (a build without BuilderFlag::kFP16, yolov8/src/model.cpp:314-324): ld in floats. */
int32_t trtx_yolo_head_decode_nhwc_f32(const void* const* heads, const int* ld, int n_levels, int batch, int classes,
                                   int net_h, int net_w, const int* strides, const float* dfl_weights, int max_out,
                                   float* output, void* workspace, size_t workspace_bytes, trtx_stream_t stream);

/*
 * Class-aware greedy NMS over the decode buffer.  Replaces host batch_nms()/nms()
 * (reference yolov8/src/postprocess.cpp:71-129; call site yolov8/yolov8_det.cpp:240).
 *   decode_out device, fp32 [batch][1 + max_out*90]  (output of trtx_yolo_decode)
 *   keep_idx   device, int32 [batch][max_out]  decode slot of every kept detection, in the reference's
 *              emission order (class ascending, conf descending, bbox[0] ascending)
 *   keep_cnt   device, int32 [batch]
 *   keep_det   device, fp32 [batch][max_out][6] = x1,y1,x2,y2,conf,class of the kept boxes; may be NULL
 *   workspace  device, >= trtx_yolo_nms_workspace(batch) bytes, 16-byte aligned (sorted records + the 1024x1024-bit
 *              suppression matrix per image)
 * max_out <= 1024.
 */
size_t trtx_yolo_nms_workspace(int batch);
int32_t trtx_yolo_nms(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh,
                      int32_t* keep_idx, int32_t* keep_cnt, float* keep_det, void* workspace, size_t workspace_bytes,
                      trtx_stream_t stream);


/*
 * Anchor-based YoloLayer (YOLOv5 / v7 / v3 / v4 family) - the reference's yolov5/plugin/yololayer.cu:161-227.
 *   inputs     n_levels device pointers, fp32 [batch][3 * (5 + classes (+ 32 if is_segmentation))][grid_h * grid_w]
 *   grid_w/h   per level (YoloKernel::width / height, yolov5/src/types.h:5-9); anchors: n_levels x 6 floats (w0,h0,w1,h1,w2,h2)
 *   output     device, fp32 [batch][1 + max_out * 38]: count (clamped to max_out), then Detection records of 38 floats
 *              (yolov5/src/types.h:11-16): cx, cy, w, h, conf = obj * cls, class_id, mask[32]
 * Candidates with sigmoid(obj) >= 0.1 (kIgnoreThresh) in canonical (level, cell, anchor) order (the reference: atomicAdd race).
 */
size_t trtx_yolov5_decode_workspace(int batch, const int* grid_w, const int* grid_h, int n_levels);
int32_t trtx_yolov5_decode(const float* const* inputs, int n_levels, int batch, int classes, int net_h, int net_w,
                           const int* grid_w, const int* grid_h, const float* anchors, int max_out, int is_segmentation,
                           float* output, void* workspace, size_t workspace_bytes, trtx_stream_t stream);
/* Oriented boxes (yolov8 obb): host nms_obb() on the GPU (yolov8/src/postprocess.cpp:303-393, ProbIoU, "conf <= thresh" dropped,
 * erased when probiou >= nms_thresh); keep_det is [batch][max_out][7] = cx, cy, w, h, conf, class, angle.  And the reference's
 * GPU mode for oriented boxes, cuda_decode_obb + cuda_nms_obb (yolov8/src/postprocess.cu:7-40, 113-166): out [batch][1 + max_out*8]. */
int32_t trtx_yolo_nms_obb(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh, int32_t* keep_idx,
                          int32_t* keep_cnt, float* keep_det, void* workspace, size_t workspace_bytes, trtx_stream_t stream);
int32_t trtx_yolo_postprocess_gpu_obb(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh, float* out,
                                      trtx_stream_t stream);
/* YOLOv5 host nms() / batch_nms() on the GPU (yolov5/src/postprocess.cpp:30-80): centre-format IoU, conf <= thresh dropped,
 * class-wise greedy suppression in conf-descending order.  Same outputs / workspace as trtx_yolo_nms; records are 38 floats. */
int32_t trtx_yolov5_nms(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh, int32_t* keep_idx,
                        int32_t* keep_cnt, float* keep_det, void* workspace, size_t workspace_bytes, trtx_stream_t stream);

/*
 * Letterbox pre-processing (the reference's cuda_preprocess / cuda_batch_preprocess, yolov8/src/preprocess.cu:7-127):
 * uint8 HWC BGR image -> fp32 CHW RGB / 255, bilinear warp-affine about the image centres, border 128.
 *   trtx_letterbox_batch   sources already on the device: src[i] device pointer, tightly packed src_w[i] x src_h[i] x 3;
 *                          dst device fp32 [batch][3][dst_h][dst_w]; ONE launch for the batch, nothing synchronises.
 *   trtx_preprocess_init / _destroy / trtx_batch_preprocess   host images (cuda_preprocess_init(max_image_size), :129-139):
 *                          pinned ring + device ring + copy stream; batch <= ring_depth.
 *   trtx_letterbox_matrix  the dst->src map the kernel uses (s2d about the centres, inverted as cv::invertAffineTransform does).
 */
void trtx_letterbox_matrix(int src_w, int src_h, int dst_w, int dst_h, float* d2s_out /* 6 floats */);
int32_t trtx_letterbox_batch(const void* const* src, const int* src_w, const int* src_h, int batch, float* dst, int dst_w, int dst_h,
                             trtx_stream_t stream);
int32_t trtx_preprocess_init(int max_image_size, int ring_depth);
void trtx_preprocess_destroy(void);
int32_t trtx_batch_preprocess(const void* const* src_host, const int* src_w, const int* src_h, int batch, float* dst, int dst_w, int dst_h,
                              trtx_stream_t stream);

/*
 * The reference's optional GPU post-processing mode "g": cuda_decode + cuda_nms (yolov8/src/postprocess.cu:42-111, call
 * site yolov8/yolov8_det.cpp:105-112; batch 1 only there, any batch here).  NOT the same result as trtx_yolo_nms: the
 * suppression is non-greedy (a box is dropped if any same-class box with higher confidence overlaps it).
 *   out  device, fp32 [batch][1 + max_out*7]: out[0] = input count, then records x1,y1,x2,y2,conf,class,keep (1/0) in input
 *        slot order; records of positions with conf < conf_thresh are all-zero.  max_out <= 1024.
 */
int32_t trtx_yolo_postprocess_gpu(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh,
                                  float* out, trtx_stream_t stream);

/* ---- Mish (yolov4, scaled-yolov4 and the early yolov5 samples) ------------------------------------ */
/* mish_kernel of MishPlugin::forwardGpu (reference yolov4/mish.cu:113-141): out[i] = in[i] * tanh(softplus(in[i])) over n device fp32
 * values, softplus with the reference's threshold of 20 (x > 20 -> x, x < -20 -> exp(x)), tanh spelled 2 / (1 + exp(-2y)) - 1. */
int32_t trtx_mish(const float* in, float* out, size_t n, trtx_stream_t stream);

/* ---- RetinaFace ------------------------------------------------------------------------------- */
/*
 * DecodePlugin::enqueue (reference retinaface/decode.cu:110-191).  inputs[l] (l = stride 8/16/32): device fp32
 * [batch][32][h_l*w_l] = bbox(2x4) | cls(2x2) | landmark(2x10) planes (retina_r50.cpp:197-202).
 * output: [batch][1 + anchors*15]: count, then Detection{bbox[4] xyxy px, conf, landmark[10]} (decode.h:11-15)
 * for every anchor with softmax conf > 0.02, in (level, cell, k) order.  The reference fixes INPUT_H/W at
 * compile time (decode.h:16-17); here they are arguments.
 */
size_t trtx_retina_decode_output_floats(int net_h, int net_w);
size_t trtx_retina_decode_workspace(int batch, int net_h, int net_w);
int32_t trtx_retina_decode(const float* const* inputs, int batch, int net_h, int net_w, float* output, void* workspace,
                           size_t workspace_bytes, trtx_stream_t stream);
/*
 * Host nms() of the reference (retinaface/common.hpp:91-130) on the device: conf > conf_thresh (0.1, compared in
 * double), sort by conf descending, greedy suppression with iou(+1e-6) > nms_thresh.  keep_idx [batch][max_keep]
 * = decode slots in emission order, keep_det (nullable) [batch][max_keep][15] the kept records.
 */
size_t trtx_retina_nms_workspace(int batch, int net_h, int net_w);
int32_t trtx_retina_nms(const float* decode_out, int batch, int net_h, int net_w, double conf_thresh, float nms_thresh,
                        int max_keep, int32_t* keep_idx, int32_t* keep_cnt, float* keep_det, void* workspace,
                        size_t workspace_bytes, trtx_stream_t stream);

/* ---- R-CNN (detectron2-style) plugin chain ------------------------------------------------------ */
/* rpnDecode (rcnn/RpnDecode.cu:27-143): stable descending sort of the A*h*w logits, top_n, anchor delta decode
 * + clip; empty boxes get score -FLT_MAX.  scores [batch][A*h*w], deltas [batch][A*4*h*w] (CHW planes),
 * anchors_host: host fp32 [A*4] (GenerateAnchors, rcnn.cpp:62-77). */
size_t trtx_rpn_decode_workspace(int batch, int num_anchors, int height, int width);
int32_t trtx_rpn_decode(int batch, const float* scores, const float* deltas, int height, int width, int image_height,
                        int image_width, float stride, const float* anchors_host, int num_anchors, int top_n,
                        float* out_scores, float* out_boxes, void* workspace, size_t workspace_bytes,
                        trtx_stream_t stream);
/* rpnNms (rcnn/RpnNms.cu:59-121): sort, exact greedy NMS (the reference kernel races across blocks), stable
 * re-sort, first post_nms_topk boxes.  scores [batch][pre], boxes [batch][pre][4] -> out_boxes [batch][post][4] */
size_t trtx_sorted_nms_workspace(int batch, int n);
int32_t trtx_rpn_nms(int batch, const float* scores, const float* boxes, int pre_nms_topk, int post_nms_topk,
                     float nms_thresh, float* out_boxes, void* workspace, size_t workspace_bytes,
                     trtx_stream_t stream);
/* roiAlign (rcnn/RoiAlign.cu:83-182): detectron2 ROIAlign(aligned=True), adaptive sampling when ratio == 0.
 * boxes [batch][P][4], features [batch][C][fh][fw] -> out [batch][P][C][res][res] */
int32_t trtx_roi_align(int batch, const float* boxes, const float* features, int pooler_resolution, float spatial_scale,
                       int sampling_ratio, int num_proposals, int channels, int feature_h, int feature_w, float* out,
                       trtx_stream_t stream);
/* The same operator on the engine's native layout: features NHWC fp16 [batch][fh][fw][ld_in] (channels % 8 == 0), out NHWC fp16
 * [batch * P][res][res][ld_out] - what the RoI head's convolutions consume.  Used by the fp16 engine in place of the fp32
 * plugin edge + two layout passes. */
int32_t trtx_roi_align_nhwc_f16(int batch, const float* boxes, const void* features, int ld_in, int pooler_resolution,
                                float spatial_scale, int sampling_ratio, int num_proposals, int channels, int feature_h,
                                int feature_w, void* out, int ld_out, trtx_stream_t stream);
/* ... computing only every bin_step-th bin per axis: out [batch * P][ores][ores][ld_out], ores = (res - 1) / bin_step + 1, bin (oh, ow) of
 * the output = bin (oh * bin_step, ow * bin_step) of the res x res grid, same arithmetic.  bin_step 2 is what res5.0's two 1x1 STRIDE-2
 * convolutions read of the 14 x 14 RoI features (rcnn/backbone.hpp:9,110-117 STRIDE_IN_1X1; rcnn/rcnn.cpp:147-163): the engine emits
 * the 7 x 7 grid and runs those convolutions at stride 1 - three quarters of the RoIAlign samples and writes are gone. */
int32_t trtx_roi_align_nhwc_f16_strided(int batch, const float* boxes, const void* features, int ld_in, int pooler_resolution,
                                        float spatial_scale, int sampling_ratio, int num_proposals, int channels, int feature_h,
                                        int feature_w, void* out, int ld_out, int bin_step, trtx_stream_t stream);
/* predictorDecode (rcnn/PredictorDecode.cu:24-110): sort N*C scores, top N (n, cls) pairs, weighted delta decode.
 * The reference clips y2 with image_width (line 99); kept for parity. */
size_t trtx_predictor_decode_workspace(int batch, int num_boxes, int num_classes);
int32_t trtx_predictor_decode(int batch, const float* scores, const float* deltas, const float* proposals, int num_boxes,
                              int num_classes, int image_height, int image_width, const float* bbox_reg_weights_host,
                              float* out_scores, float* out_boxes, float* out_classes, void* workspace,
                              size_t workspace_bytes, trtx_stream_t stream);
/* batchedNms (rcnn/BatchedNms.cu:28-162): class-aware; method 0 hard, 1 soft-NMS linear (reference default,
 * rcnn.cpp:59), 2 soft-NMS gaussian; workspace: trtx_sorted_nms_workspace(batch, count) */
int32_t trtx_batched_nms(int nms_method, int batch, const float* scores, const float* boxes, const float* classes,
                         int count, int detections_per_im, float nms_thresh, float* out_scores, float* out_boxes,
                         float* out_classes, void* workspace, size_t workspace_bytes, trtx_stream_t stream);

/* maskRcnnInference (rcnn/MaskRcnnInference.cu:8-62): sigmoid of the mask plane of each detection's predicted class.
 * labels [batch][D] (class ids as floats), masks [batch][D][C][S][S] -> out_masks [batch][D][1][S][S]; class ids outside
 * [0, C) give a zero plane (the reference leaves it unwritten). */
int32_t trtx_mask_rcnn_inference(int batch, const float* labels, const float* masks, int detections_per_im, int output_size,
                                 int num_classes, float* out_masks, trtx_stream_t stream);

/* ---- single-kernel entry points (parity tests / micro-benchmarks) ----------------------------- */
/* activation codes for the fused conv epilogue */
#define TRTX_ACT_NONE 0
#define TRTX_ACT_RELU 1
#define TRTX_ACT_SIGMOID 2
#define TRTX_ACT_SILU 3
#define TRTX_ACT_LEAKY 4
#define TRTX_ACT_TANH 5
/* dims of the packed fp16 weight image [cout_pad][kpad] the implicit-GEMM kernel reads */
int32_t trtx_conv_packed_dims(int cout, int cin_pad, int kh, int kw, int32_t* cout_pad, int32_t* kpad, int32_t* bn);
/* host: KCRS fp32 -> packed fp16 (optional per-output-channel scale = folded BatchNorm) */
int32_t trtx_conv_pack_weights_f16(const float* w_kcrs, int cout, int cin, int kh, int kw, int cin_pad,
                                   const float* ch_scale, uint16_t* packed);
/* one fused convolution launch: NHWC fp16 in/out (channel strides ld_*), y = act2(act1(conv+bias) + residual) */
int32_t trtx_op_conv2d_nhwc_f16(const void* in, int N, int H, int W, int Cin, int ld_in, const void* wpacked,
                                const float* bias, void* out, int Cout, int ld_out, int kh, int kw, int sh, int sw,
                                int ph, int pw, int act1, const void* residual, int ld_res, int act2,
                                trtx_stream_t stream);
/* Tactics of that launch (tests / tools): the exchangeable launch configurations of the layer trtx_op_conv2d_nhwc_f16 would
 * run, 6 ints each {column-tile width, k-step width, rows per tile, wave-split-K (1 off / 2 on), weight-stationary (1 off / 2 on),
 * 3x3 row-reuse kernel (0 no / 1 three LDS stages / 2 two)}; entry 0 is the default.  Returns the count (<= max_out).  trtx_op_conv_force_tactic pins the configuration used by the following
 * trtx_op_conv2d_nhwc_f16 calls of this process (NULL: back to the default). */
int32_t trtx_op_conv2d_tactics(int N, int H, int W, int Cin, int ld_in, int Cout, int ld_out, int kh, int kw, int sh, int sw, int ph, int pw,
                               int has_residual, int ld_res, int32_t* out6, int32_t max_out);
int32_t trtx_op_conv_force_tactic(const int32_t* tactic6);
/* The same launch in an fp32 engine (builds without BuilderFlag::kFP16: yolov8/include/config.h:1-3 USE_FP32, yolov8/src/model.cpp:314-324):
 * NHWC fp32 in / out / residual, fp32 weights packed [cout_pad][kpad] (k = tap * cink + c, cink = Cin rounded up to the 16-channel k-step, or 8 for
 * Cin <= 8; Cin itself a multiple of 4), fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products and sums).  trtx_op_conv2d_tactics_f32 lists the tile
 * configurations of the layer, 4 ints each {column-tile width, rows per tile, operand path: 1 = LDS-DMA, 3 = resident patch, 5 = through registers, 6 = wave roles,
 * 7 / 8 = the resident-operand 3x3 / 1x1 kernels of conv_res.hip, channels per k-step: 16 / 32} - entry 0 the launcher's own choice, every entry the same bits -;
 * tile4 pins one for this call (NULL: the launcher's choice).  `bias` (may be NULL) holds cout_pad floats (trtx_conv_packed_dims_f32; Cout rounded up to 16), 16-byte
 * aligned: the tile starts its accumulators with 16-byte loads of it up to the padded width - a [Cout] array with Cout % 16 != 0 would be read past its end. */
int32_t trtx_conv_packed_dims_f32(int cout, int cin_pad, int kh, int kw, int32_t* cout_pad, int32_t* kpad, int32_t* cink);
int32_t trtx_conv_pack_weights_f32(const float* w_kcrs, int cout, int cin, int kh, int kw, int cin_pad, const float* ch_scale, float* packed);
int32_t trtx_op_conv2d_nhwc_f32(const void* in, int N, int H, int W, int Cin, int ld_in, const void* wpacked, const float* bias, void* out, int Cout,
                                int ld_out, int kh, int kw, int sh, int sw, int ph, int pw, int act1, const void* residual, int ld_res, int act2,
                                const int32_t* tile4, trtx_stream_t stream);
int32_t trtx_op_conv2d_tactics_f32(int N, int H, int W, int Cin, int ld_in, int Cout, int ld_out, int kh, int kw, int sh, int sw, int ph, int pw,
                                   int has_residual, int ld_res, int32_t* out4, int32_t max_out);
/* kINT8 convolution, single-kernel entries for the parity tests (what TensorRT runs for a layer of a USE_INT8 build, yolov8/include/config.h:1-3): KCRS fp32 weights ->
 * int8 [cout_pad][kpad bytes] with per-output-channel symmetric scales (cink = Cin rounded up to 64 channels, kpad = kh*kw*cink; packed / wscale_out NULL: sizes only);
 * the int8 MFMA convolution on NHWC int8 input: out int8 (out_is_i8, quantised with out_inv_scale) or fp16, cscale[cout_pad] = input scale x weight scale, optional
 * int8 / fp16 shortcut. */
int32_t trtx_conv_pack_weights_i8(const float* w_kcrs, int cout, int cin, int kh, int kw, const float* ch_scale, int8_t* packed, float* wscale_out,
                                  int32_t* cout_pad_out, int32_t* kpad_out);
int32_t trtx_op_conv2d_nhwc_i8(const void* in, int N, int H, int W, int Cin, int ld_in, const void* wpacked, const float* cscale, const float* bias, void* out,
                               int out_is_i8, float out_inv_scale, int Cout, int ld_out, int kh, int kw, int sh, int sw, int ph, int pw, int act1,
                               const void* residual, int res_is_i8, float res_scale, int ld_res, int act2, trtx_stream_t stream);
int32_t trtx_op_poison_lds(void* device_word, trtx_stream_t stream); /* test support: NaN patterns into every CU's LDS */
int32_t trtx_op_nchw_f32_to_nhwc_f16(const float* in, void* out, int N, int C, int H, int W, int Cpad, int ld_out,
                                     trtx_stream_t stream);
int32_t trtx_op_nhwc_f16_to_nchw_f32(const void* in, float* out, int N, int C, int H, int W, int ld_in,
                                     trtx_stream_t stream);

/* ================= Section 2: network definition / engine / execution context =================== */
/*
 * Mirrors, object for object, the nvinfer1 API subset the reference builders use (SURVEY.md §2.3,
 * §8b): IBuilder / IBuilderConfig / INetworkDefinition / ITensor / ILayer / IHostMemory / IRuntime /
 * ICudaEngine / IExecutionContext / IPluginV2 / IPluginCreator / getPluginRegistry().
 * include/NvInfer.h is the header-only C++ shim that gives these the nvinfer1 spelling.
 *
 * Ownership: layers and tensors are owned by their network (never freed by the caller); weights are
 * COPIED when a layer is added, so the caller may free its host blobs after build (the reference frees
 * them after buildSerializedNetwork, yolov8/src/model.cpp:332-334).
 */
typedef struct trtx_builder trtx_builder;
typedef struct trtx_network trtx_network;
typedef struct trtx_hostmem trtx_hostmem;
typedef struct trtx_engine trtx_engine;
typedef struct trtx_context trtx_context;
typedef struct trtx_plugin trtx_plugin; /* runtime-side handle of one plugin instance */

#define TRTX_MAX_DIMS 8
typedef struct trtx_dims {
    int32_t nb;
    int64_t d[TRTX_MAX_DIMS];
} trtx_dims;

/* nvinfer1::DataType */
#define TRTX_DTYPE_FLOAT 0
#define TRTX_DTYPE_HALF 1
#define TRTX_DTYPE_INT8 2
#define TRTX_DTYPE_INT32 3
/* nvinfer1::ActivationType */
#define TRTX_ACTIVATION_RELU 0
#define TRTX_ACTIVATION_SIGMOID 1
#define TRTX_ACTIVATION_TANH 2
#define TRTX_ACTIVATION_LEAKY_RELU 3
/* nvinfer1::PoolingType */
#define TRTX_POOLING_MAX 0
#define TRTX_POOLING_AVERAGE 1
/* nvinfer1::ElementWiseOperation */
#define TRTX_ELEMENTWISE_SUM 0
#define TRTX_ELEMENTWISE_PROD 1
#define TRTX_ELEMENTWISE_MAX 2
#define TRTX_ELEMENTWISE_MIN 3
#define TRTX_ELEMENTWISE_SUB 4
#define TRTX_ELEMENTWISE_DIV 5
#define TRTX_ELEMENTWISE_POW 6
/* nvinfer1::ScaleMode */
#define TRTX_SCALE_UNIFORM 0
#define TRTX_SCALE_CHANNEL 1
#define TRTX_SCALE_ELEMENTWISE 2
/* nvinfer1::ReduceOperation */
#define TRTX_REDUCE_SUM 0
#define TRTX_REDUCE_PROD 1
#define TRTX_REDUCE_MAX 2
#define TRTX_REDUCE_MIN 3
#define TRTX_REDUCE_AVG 4
/* nvinfer1::MatrixOperation */
#define TRTX_MATMUL_NONE 0
#define TRTX_MATMUL_TRANSPOSE 1
#define TRTX_MATMUL_VECTOR 2
/* nvinfer1::ResizeMode */
#define TRTX_RESIZE_NEAREST 0
#define TRTX_RESIZE_LINEAR 1
/* nvinfer1::BuilderFlag */
#define TRTX_FLAG_FP16 0
#define TRTX_FLAG_INT8 1
/* layer parameter ids for trtx_layer_set_ints / _floats / _dims */
#define TRTX_P_STRIDE 1
#define TRTX_P_PADDING 2
#define TRTX_P_DILATION 3
#define TRTX_P_GROUPS 4
#define TRTX_P_ALPHA 5
#define TRTX_P_BETA 6
#define TRTX_P_AXIS 7          /* concat axis; softmax / reduce axes bitmask */
#define TRTX_P_RESHAPE 8       /* IShuffleLayer::setReshapeDimensions */
#define TRTX_P_FIRST_TRANSPOSE 9
#define TRTX_P_SECOND_TRANSPOSE 10
#define TRTX_P_RESIZE_MODE 11
#define TRTX_P_RESIZE_SCALES 12
#define TRTX_P_RESIZE_OUT_DIMS 13
#define TRTX_P_AVG_EXCLUSIVE 14
#define TRTX_P_KERNEL 15
#define TRTX_P_NB_OUT 16

/* --- plugins: C v-table an IPluginV2-style object is driven through --------------------------- */
/* Every callback receives `self`.  Semantics and call order follow IPluginV2Ext/IOExt as the reference
 * uses them (yolov8/plugin/yololayer.h:7-81; rcnn/RpnDecodePlugin.h:29-187; SURVEY.md §8b lifecycle):
 * clone at addPluginV2 -> configure -> workspace_size -> initialize -> serialize at plan write;
 * on load: creator.deserialize -> configure -> initialize -> enqueue xN -> terminate -> destroy.
 * Tensors at the plugin edge are fp32 LINEAR (NCHW), as supportsFormatCombination demands in the
 * reference (yololayer.h:32-35). */
typedef struct trtx_plugin_vtbl {
    void* self;
    int32_t (*get_nb_outputs)(void* self);
    int32_t (*get_output_dims)(void* self, int32_t index, const trtx_dims* inputs, int32_t nb_inputs, trtx_dims* out);
    int32_t (*configure)(void* self, const trtx_dims* in, int32_t nb_in, const trtx_dims* out, int32_t nb_out,
                         int32_t max_batch);
    int32_t (*initialize)(void* self);
    void (*terminate)(void* self);
    size_t (*workspace_size)(void* self, int32_t max_batch);
    int32_t (*enqueue)(void* self, int32_t batch, const void* const* inputs, void* const* outputs, void* workspace,
                       trtx_stream_t stream); /* 0 = OK, as IPluginV2::enqueue */
    size_t (*serialization_size)(void* self);
    void (*serialize)(void* self, void* buffer);
    const char* (*plugin_type)(void* self);
    const char* (*plugin_version)(void* self);
    /* fills *out with a v-table for an independent copy; returns 0 on success */
    int32_t (*clone)(void* self, struct trtx_plugin_vtbl* out);
    void (*destroy)(void* self);
} trtx_plugin_vtbl;

typedef struct trtx_plugin_field {
    const char* name;
    const void* data;
    int32_t type; /* nvinfer1::PluginFieldType: 0 f16, 1 f32, 2 f64, 3 i8, 4 i16, 5 i32, 6 char */
    int32_t length;
} trtx_plugin_field;

typedef struct trtx_creator_vtbl {
    void* self;
    const char* (*plugin_name)(void* self);
    const char* (*plugin_version)(void* self);
    int32_t (*create)(void* self, const char* name, const trtx_plugin_field* fields, int32_t nb_fields,
                      trtx_plugin_vtbl* out);
    int32_t (*deserialize)(void* self, const char* name, const void* data, size_t length, trtx_plugin_vtbl* out);
} trtx_creator_vtbl;

/* getPluginRegistry()->registerCreator / getPluginCreator (REGISTER_TENSORRT_PLUGIN, yololayer.h:109).
 * Built-in HIP plugins are pre-registered: "YoloLayer_TRT"/"1", "Decode_TRT"/"1" (retinaface). */
int32_t trtx_registry_register(const trtx_creator_vtbl* creator);
int32_t trtx_registry_get(const char* name, const char* version, trtx_creator_vtbl* out);

/* --- .wts weight files (loadWeights, lenet/utils.h:49-80; yolov8/src/block.cpp:13-43) ----------- */
typedef struct trtx_wts trtx_wts;
int32_t trtx_wts_load(const char* path, trtx_wts** out);
int32_t trtx_wts_count(const trtx_wts* w);
/* i-th entry in file order; values are the host fp32 array owned by `w` */
int32_t trtx_wts_entry(const trtx_wts* w, int32_t i, const char** name, const float** values, int64_t* count);
int32_t trtx_wts_find(const trtx_wts* w, const char* name, const float** values, int64_t* count);
void trtx_wts_free(trtx_wts* w);

/* --- builder / config (createInferBuilder, IBuilderConfig) --------------------------------------- */
int32_t trtx_builder_create(trtx_builder** out);
void trtx_builder_destroy(trtx_builder* b);
int32_t trtx_builder_set_max_batch(trtx_builder* b, int32_t n);        /* IBuilder::setMaxBatchSize */
int32_t trtx_builder_set_flag(trtx_builder* b, int32_t flag, int32_t on); /* IBuilderConfig::setFlag */
int32_t trtx_builder_set_workspace(trtx_builder* b, size_t bytes);     /* setMaxWorkspaceSize / setMemoryPoolLimit */
/* IBuilderConfig::setInt8Calibrator (yolov8/src/model.cpp:317-324): the calibrator as a C v-table, IInt8EntropyCalibrator2's
 * methods (yolov8/include/calibrator.h:14-36).  With BuilderFlag::kINT8, trtx_build_serialized first asks read_cache; if that
 * yields a cache the scales come from it (no GPU needed), otherwise get_batch is called until it returns 0, every batch runs
 * through an fp16 engine of the network on the GPU, per-tensor |x| histograms are reduced to thresholds by entropy minimisation
 * and the resulting cache text is handed to write_cache. */
typedef struct trtx_calibrator_vtbl {
    void* self;
    int32_t (*get_batch_size)(void* self);
    /* fills bindings[i] with the DEVICE pointer of input names[i] for one batch; returns 1, or 0 when the data is exhausted */
    int32_t (*get_batch)(void* self, void** bindings, const char* const* names, int32_t nb_bindings);
    const void* (*read_cache)(void* self, size_t* length); /* NULL / length 0: no cache */
    void (*write_cache)(void* self, const void* cache, size_t length);
    /* IInt8Calibrator::getAlgorithm (nvinfer1::CalibrationAlgoType): NULL or 1 / 2 = entropy calibration (the reference's
     * IInt8EntropyCalibrator2: the threshold that minimises the KL divergence, clipping rare large values); 3 = kMINMAX_CALIBRATION
     * (IInt8MinMaxCalibrator: the largest |x| seen, nothing clipped). */
    int32_t (*get_algorithm)(void* self);
} trtx_calibrator_vtbl;
/* IBuilderConfig::setMaxAuxStreams (TensorRT >= 8.6): how many streams besides the caller's an execution context may use to run
 * independent branches of the plan concurrently.  -1 (default) = the runtime's choice (3); 0 = strictly the caller's stream, the
 * right setting when several execution contexts are kept in flight on their own streams (bench.py --contexts). */
int32_t trtx_builder_set_max_aux_streams(trtx_builder* b, int32_t n);
int32_t trtx_builder_set_int8_calibrator(trtx_builder* b, const trtx_calibrator_vtbl* calibrator);
/* the threshold search of the entropy calibration on a caller-supplied histogram of |x| over [0, range] (tests, tools) */
float trtx_int8_entropy_threshold(const double* hist, int32_t bins, float range);
/* the clip limit the calibration applies on top of it (round 6; TRTX_INT8_CLIP_LIMIT, default 1e-4): the lowest bin edge >= thr beyond which at most `limit` of the
 * histogram's mass lies; limit <= 0 returns thr.  TensorRT's entropy calibrator is closed: both searches are this library's own. */
float trtx_int8_clip_limited_threshold(const double* hist, int32_t bins, float range, float thr, double limit);
/* createNetworkV2(flags): bit 0 = kEXPLICIT_BATCH */
int32_t trtx_network_create(trtx_builder* b, uint32_t flags, trtx_network** out);
void trtx_network_destroy(trtx_network* n);
const char* trtx_network_last_error(const trtx_network* n);

/* --- INetworkDefinition::add*: return a layer index (>= 0) or -1; outputs via trtx_layer_output ---- */
int32_t trtx_add_input(trtx_network* n, const char* name, int32_t dtype, const trtx_dims* dims); /* tensor id */
int32_t trtx_add_convolution(trtx_network* n, int32_t input, int32_t nb_out, int32_t kh, int32_t kw,
                             const float* kernel, int64_t kernel_count, const float* bias, int64_t bias_count);
int32_t trtx_add_deconvolution(trtx_network* n, int32_t input, int32_t nb_out, int32_t kh, int32_t kw,
                               const float* kernel, int64_t kernel_count, const float* bias, int64_t bias_count);
int32_t trtx_add_fully_connected(trtx_network* n, int32_t input, int32_t nb_out, const float* kernel,
                                 int64_t kernel_count, const float* bias, int64_t bias_count);
int32_t trtx_add_activation(trtx_network* n, int32_t input, int32_t type);
int32_t trtx_add_pooling(trtx_network* n, int32_t input, int32_t type, int32_t kh, int32_t kw);
int32_t trtx_add_scale(trtx_network* n, int32_t input, int32_t mode, const float* shift, int64_t shift_count,
                       const float* scale, int64_t scale_count, const float* power, int64_t power_count);
int32_t trtx_add_elementwise(trtx_network* n, int32_t a, int32_t b, int32_t op);
int32_t trtx_add_concatenation(trtx_network* n, const int32_t* inputs, int32_t nb_inputs);
int32_t trtx_add_slice(trtx_network* n, int32_t input, const trtx_dims* start, const trtx_dims* size,
                       const trtx_dims* stride);
int32_t trtx_add_shuffle(trtx_network* n, int32_t input);
int32_t trtx_add_resize(trtx_network* n, int32_t input);
int32_t trtx_add_softmax(trtx_network* n, int32_t input);
int32_t trtx_add_matrix_multiply(trtx_network* n, int32_t a, int32_t op_a, int32_t b, int32_t op_b);
int32_t trtx_add_constant(trtx_network* n, const trtx_dims* dims, const float* values, int64_t count);
int32_t trtx_add_reduce(trtx_network* n, int32_t input, int32_t op, uint32_t axes, int32_t keep_dims);
int32_t trtx_add_identity(trtx_network* n, int32_t input);
/* addPluginV2: the runtime clones the plugin through vtbl->clone (as TensorRT does) */
int32_t trtx_add_plugin_v2(trtx_network* n, const int32_t* inputs, int32_t nb_inputs, const trtx_plugin_vtbl* plugin);

/* ILayer / ITensor accessors */
int32_t trtx_layer_nb_outputs(const trtx_network* n, int32_t layer);
int32_t trtx_layer_output(const trtx_network* n, int32_t layer, int32_t index); /* tensor id */
int32_t trtx_layer_set_name(trtx_network* n, int32_t layer, const char* name);
int32_t trtx_layer_set_ints(trtx_network* n, int32_t layer, int32_t param, const int32_t* v, int32_t count);
int32_t trtx_layer_set_floats(trtx_network* n, int32_t layer, int32_t param, const float* v, int32_t count);
int32_t trtx_layer_set_dims(trtx_network* n, int32_t layer, int32_t param, const trtx_dims* d);
int32_t trtx_tensor_get_dims(const trtx_network* n, int32_t tensor, trtx_dims* out);
int32_t trtx_tensor_set_name(trtx_network* n, int32_t tensor, const char* name);
const char* trtx_tensor_get_name(const trtx_network* n, int32_t tensor);
int32_t trtx_mark_output(trtx_network* n, int32_t tensor);

/* IBuilder::buildSerializedNetwork -> IHostMemory.  Works without a GPU (pure host work). */
int32_t trtx_build_serialized(trtx_builder* b, trtx_network* n, trtx_hostmem** out);
const void* trtx_hostmem_data(const trtx_hostmem* m);
size_t trtx_hostmem_size(const trtx_hostmem* m);
void trtx_hostmem_destroy(trtx_hostmem* m);
/* a host-memory object holding a copy of `size` bytes (the shim's ICudaEngine::serialize on a machine where engines are only built) */
int32_t trtx_hostmem_create(const void* data, size_t size, trtx_hostmem** out);
/* JSON description of a serialized plan: network definition with weight offsets into the plan
 * (used by the test oracle's graph interpreter) and, with lowered != 0, the fused kernel schedule,
 * buffer plan and FLOP/byte counts the engine would execute.  Caller frees with trtx_string_free. */
int32_t trtx_plan_describe(const void* plan, size_t size, int32_t lowered, char** json_out);
void trtx_string_free(char* s);

/* --- IRuntime / ICudaEngine / IExecutionContext ---------------------------------------------------- */
/* deserializeCudaEngine: needs a gfx950 device (TRTX_ERR_NO_DEVICE otherwise — no CPU fallback) */
int32_t trtx_engine_deserialize(const void* plan, size_t size, trtx_engine** out);
int32_t trtx_engine_serialize(const trtx_engine* e, trtx_hostmem** out);
void trtx_engine_destroy(trtx_engine* e);
int32_t trtx_engine_nb_bindings(const trtx_engine* e);
int32_t trtx_engine_binding_index(const trtx_engine* e, const char* name);
const char* trtx_engine_binding_name(const trtx_engine* e, int32_t index);
int32_t trtx_engine_binding_is_input(const trtx_engine* e, int32_t index);
int32_t trtx_engine_binding_dims(const trtx_engine* e, int32_t index, trtx_dims* out);
int32_t trtx_engine_binding_dtype(const trtx_engine* e, int32_t index);
int32_t trtx_engine_max_batch(const trtx_engine* e);
size_t trtx_engine_device_memory(const trtx_engine* e);
/* Tactic selection (TensorRT's builder times several kernels per layer and keeps the fastest; buildSerializedNetwork,
 * yolov8/src/model.cpp:327).  Here it runs inside trtx_engine_deserialize: every MFMA convolution's exchangeable launch
 * configurations are timed in place, behind their real producers, and a layer leaves its default for one that is >= 3 % faster
 * (TRTX_TUNE=0 in the environment at deserialize: every layer keeps its static default and this returns []; TRTX_TACTIC_CACHE=<file>: choices are read from / appended to that file, the ITimingCache
 * analogue, so a later process does not time again).  Returns JSON [{op, name, tactic, default, us, default_us, candidates}, ...]; free with
 * trtx_string_free. */
int32_t trtx_engine_tactics(const trtx_engine* e, char** json_out);
/* Multi-GPU in one process (reference: tutorials/multi_GPU_processing.md:13-30, one `Plan` per device after cudaSetDevice(i)):
 * an engine is bound to the HIP device that was current at trtx_engine_deserialize; this returns its ordinal (-1 for NULL).
 * trtx_context_create / enqueue / enqueue_v3 / profile return TRTX_ERR_STATE when another device is current. */
int32_t trtx_engine_device(const trtx_engine* e);
int32_t trtx_context_create(trtx_engine* e, trtx_context** out);
void trtx_context_destroy(trtx_context* c);
/* IExecutionContext::enqueue(batch, bindings, stream, nullptr) — implicit batch */
int32_t trtx_context_enqueue(trtx_context* c, int32_t batch, void* const* bindings, trtx_stream_t stream);
/* cuda_batch_preprocess + enqueue in one call (yolov8/yolov8_det.cpp:146-160 -> preprocess.cu:119-127): frames[i] is a DEVICE pointer to a
 * tightly packed uint8 HWC BGR image of frame_w[i] x frame_h[i]; the engine's first layer samples the letterboxed frame itself, the fp32
 * network input is never materialised (the input binding's pointer is ignored and may be NULL).  Same output bits as
 * trtx_letterbox_batch followed by trtx_context_enqueue.  Engines whose first layer is a 3-channel stem convolution on the MFMA path
 * (every YOLO / RetinaFace / ResNet builder of the reference); TRTX_ERR_UNSUPPORTED otherwise - then use the two calls. */
int32_t trtx_context_enqueue_frames(trtx_context* c, int32_t batch, const void* const* frames, const int32_t* frame_w, const int32_t* frame_h,
                                    void* const* bindings, trtx_stream_t stream);
/* setTensorAddress + enqueueV3 — explicit batch */
int32_t trtx_context_set_tensor_address(trtx_context* c, const char* name, void* ptr);
int32_t trtx_context_enqueue_v3(trtx_context* c, trtx_stream_t stream);
/* per-kernel timing of the last profiled enqueue (IProfiler::reportLayerTime analogue):
 * runs one enqueue with hipEvents around every launch and returns JSON [{name, kind, ms}, ...] */
int32_t trtx_context_profile(trtx_context* c, int32_t batch, void* const* bindings, trtx_stream_t stream,
                             char** json_out);

#ifdef __cplusplus
}
#endif
#endif /* TRTX_HIP_H_ */
