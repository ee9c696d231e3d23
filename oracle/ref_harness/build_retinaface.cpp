// ORACLE — test infrastructure only.  The reference's RetinaFace-R50 program (retinaface/retina_r50.cpp: createEngine :101-242,
// APIToModel :244-262) compiled unmodified except for its documented precision switch (`#define USE_INT8  // set USE_INT8 or USE_FP16
// or USE_FP32`, :12, set to USE_FP16 by oracle/ref_build.py: the INT8 build needs the WIDER-face calibration images);
// ref_build_retinaface runs ITS APIToModel from run_dir (where "../retinaface.wts" must resolve).  Input size = the reference's
// compile-time decodeplugin::INPUT_H x INPUT_W (decode.h:16-17: 480 x 640).
#define main ref_main_retinaface
#include "retina_r50.cpp"
#undef main
#include "build_include_main.h"

REF_EXPORT int ref_build_retinaface(const char* run_dir, int max_batch, void** out, size_t* len) {
    if (chdir(run_dir) != 0) return 2;
    nvinfer1::IHostMemory* m = nullptr;
    APIToModel((unsigned)max_batch, &m);
    return ref_copy_out(m, out, len);
}
REF_EXPORT int ref_retinaface_input_h(void) { return INPUT_H; }
REF_EXPORT int ref_retinaface_input_w(void) { return INPUT_W; }
