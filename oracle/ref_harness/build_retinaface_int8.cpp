// ORACLE — test infrastructure only.  The reference's RetinaFace-R50 program in its DEFAULT configuration (retinaface/retina_r50.cpp:12
// `#define USE_INT8`, :219-225: BuilderFlag::kINT8 + ITS Int8EntropyCalibrator2 from retinaface/calibrator.cpp), compiled unmodified.
// ref_build_retinaface_int8 runs ITS APIToModel from run_dir, where "../retinaface.wts" and the calibration table the program names
// ("r50_int8calib.table") must exist: the calibrator then hands the builder the cached scales and no image is read (OpenCV is a
// compile-only stand-in here) nor any GPU touched.
#define main ref_main_retinaface
#include "retina_r50.cpp"
#undef main
#include "build_include_main.h"

REF_EXPORT int ref_build_retinaface_int8(const char* run_dir, int max_batch, void** out, size_t* len) {
    if (chdir(run_dir) != 0) return 2;
    nvinfer1::IHostMemory* m = nullptr;
    APIToModel((unsigned)max_batch, &m);
    return ref_copy_out(m, out, len);
}
