// ORACLE — test infrastructure only.  The reference's Faster / Mask R-CNN program (rcnn/rcnn.cpp: DataPreprocess :80-100, RPN :102-145,
// ROIHeads :147-233, createEngine_rcnn :250-308, BuildRcnnModel :310-326; rcnn/backbone.hpp) compiled unmodified; ref_build_rcnn runs ITS
// calculateSize() (:349-366: the 480 x 640 source image of the file becomes an 800 x 1067 network input) and ITS BuildRcnnModel.
#define main ref_main_rcnn
#include "rcnn.cpp"
#undef main
#include "build_include_main.h"

REF_EXPORT int ref_build_rcnn(const char* wts_path, int max_batch, const char* precision, int mask_on, void** out, size_t* len) {
    MASK_ON = mask_on != 0;
    calculateSize();
    nvinfer1::IHostMemory* m = nullptr;
    BuildRcnnModel((unsigned)max_batch, &m, wts_path, precision);
    return ref_copy_out(m, out, len);
}
REF_EXPORT int ref_rcnn_input_h(void) { return INPUT_H; }
REF_EXPORT int ref_rcnn_input_w(void) { return INPUT_W; }
