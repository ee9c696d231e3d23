// ORACLE — test infrastructure only.  C entry points around the reference's OWN host post-processing code: the
// `.inc` files are cut out of /root/reference at build time by oracle/ref_build.py (regions listed there) and are not
// tracked.  Each family lives in its own namespace because all three define `iou`, `cmp` and `nms`.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>
#include <math.h>

namespace ref_yolov8 {
#include "yolov8/include/types.h"  // the reference's Detection (90 floats), read in place from /root/reference
#include "yolov8_nms.inc"          // yolov8/src/postprocess.cpp: iou, cmp, nms, batch_nms
#include "yolov8_nms_obb.inc"      // yolov8/src/postprocess.cpp: convariance_matrix, probiou, nms_obb, batch_nms_obb
}  // namespace ref_yolov8

namespace ref_yolov5 {
#include "yolov5/src/types.h"      // Detection (38 floats), kMaxNumOutputBbox via config.h
#include "yolov5_nms.inc"          // yolov5/src/postprocess.cpp: iou (xywh), cmp, nms, batch_nms
}  // namespace ref_yolov5

namespace ref_retina {
#include "retina_types.inc"        // retinaface/decode.h: decodeplugin::Detection (15 floats)
#include "retina_nms.inc"          // retinaface/common.hpp: iou (+1e-6), cmp, nms
}  // namespace ref_retina

template <typename D>
static int copy_out(const std::vector<D>& res, float* out, int cap) {
    const int n = (int)res.size() < cap ? (int)res.size() : cap;
    if (n) memcpy(out, res.data(), (size_t)n * sizeof(D));
    return (int)res.size();
}

extern "C" {
int ref_yolov8_det_floats() { return (int)(sizeof(ref_yolov8::Detection) / sizeof(float)); }
int ref_yolov5_det_floats() { return (int)(sizeof(ref_yolov5::Detection) / sizeof(float)); }
int ref_retina_det_floats() { return (int)(sizeof(ref_retina::decodeplugin::Detection) / sizeof(float)); }

// nms(res, output, conf_thresh, nms_thresh) on ONE image's decode buffer; kept detections in emission order
int ref_yolov8_nms(float* output, float conf_thresh, float nms_thresh, float* out, int cap) {
    std::vector<ref_yolov8::Detection> res;
    ref_yolov8::nms(res, output, conf_thresh, nms_thresh);
    return copy_out(res, out, cap);
}
// batch_nms over `batch` buffers of `output_size` floats; counts[b] = kept per image, out[b][cap][det]
void ref_yolov8_batch_nms(float* output, int batch, int output_size, float conf_thresh, float nms_thresh, float* out, int cap, int* counts) {
    std::vector<std::vector<ref_yolov8::Detection>> res;
    ref_yolov8::batch_nms(res, output, batch, output_size, conf_thresh, nms_thresh);
    const size_t det = sizeof(ref_yolov8::Detection) / sizeof(float);
    for (int b = 0; b < batch; ++b) counts[b] = copy_out(res[b], out + (size_t)b * cap * det, cap);
}
int ref_yolov8_nms_obb(float* output, float conf_thresh, float nms_thresh, float* out, int cap) {
    std::vector<ref_yolov8::Detection> res;
    ref_yolov8::nms_obb(res, output, conf_thresh, nms_thresh);
    return copy_out(res, out, cap);
}
int ref_yolov5_nms(float* output, float conf_thresh, float nms_thresh, float* out, int cap) {
    std::vector<ref_yolov5::Detection> res;
    ref_yolov5::nms(res, output, conf_thresh, nms_thresh);
    return copy_out(res, out, cap);
}
int ref_retina_nms(float* output, float nms_thresh, float* out, int cap) {
    std::vector<ref_retina::decodeplugin::Detection> res;
    ref_retina::nms(res, output, nms_thresh);
    return copy_out(res, out, cap);
}
}
