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

/*
 * Class-aware greedy NMS over the decode buffer.  Replaces host batch_nms()/nms()
 * (reference yolov8/src/postprocess.cpp:71-129; call site yolov8/yolov8_det.cpp:240).
 *   decode_out device, fp32 [batch][1 + max_out*90]  (output of trtx_yolo_decode)
 *   keep_idx   device, int32 [batch][max_out]  decode slot of every kept detection, in the reference's
 *              emission order (class ascending, conf descending, bbox[0] ascending)
 *   keep_cnt   device, int32 [batch]
 *   keep_det   device, fp32 [batch][max_out][6] = x1,y1,x2,y2,conf,class of the kept boxes; may be NULL
 * max_out <= 1024.
 */
int32_t trtx_yolo_nms(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh,
                      int32_t* keep_idx, int32_t* keep_cnt, float* keep_det, trtx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TRTX_HIP_H_ */
