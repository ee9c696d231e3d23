// ORACLE — test infrastructure only.  Exported C entry points around the reference's plain (non-plugin) CUDA functions
// of yolov8/src/postprocess.cu and preprocess.cu, which oracle/ref_build.py compiles (unmodified) into the same library:
// the library is built with hidden visibility, so these wrappers are what the tests can call.
#include "postprocess.h"
#include "preprocess.h"

#define REF_API extern "C" __attribute__((visibility("default")))

// yolov8_det.cpp:101-106 (mode "g"): memset, cuda_decode, cuda_nms on one image's decode buffer
REF_API void ref_yolov8_gpu_postprocess(float* decode_dev, int model_bboxes, float conf_thresh, float nms_thresh, float* parray_dev,
                                        int max_objects, void* stream) {
    cudaStream_t s = static_cast<cudaStream_t>(stream);
    cudaMemsetAsync(parray_dev, 0, sizeof(float) * (1 + max_objects * bbox_element), s);
    cuda_decode(decode_dev, model_bboxes, conf_thresh, parray_dev, max_objects, s);
    cuda_nms(parray_dev, nms_thresh, max_objects, s);
}

// yolov8_obb.cpp mode "g": memset, cuda_decode_obb, cuda_nms_obb
REF_API void ref_yolov8_gpu_postprocess_obb(float* decode_dev, int model_bboxes, float conf_thresh, float nms_thresh, float* parray_dev,
                                            int max_objects, void* stream) {
    cudaStream_t s = static_cast<cudaStream_t>(stream);
    cudaMemsetAsync(parray_dev, 0, sizeof(float) * (1 + max_objects * 8), s);
    cuda_decode_obb(decode_dev, model_bboxes, conf_thresh, parray_dev, max_objects, s);
    cuda_nms_obb(parray_dev, nms_thresh, max_objects, s);
}

// yolov8/src/preprocess.cu:89-117 + init/destroy: host BGR uint8 image -> device CHW fp32 RGB /255 letterboxed
REF_API void ref_yolov8_preprocess_init(int max_image_size) { cuda_preprocess_init(max_image_size); }
REF_API void ref_yolov8_preprocess_destroy() { cuda_preprocess_destroy(); }
REF_API void ref_yolov8_preprocess(unsigned char* src_host, int src_w, int src_h, float* dst_dev, int dst_w, int dst_h, void* stream) {
    cuda_preprocess(src_host, src_w, src_h, dst_dev, dst_w, dst_h, static_cast<cudaStream_t>(stream));
}
