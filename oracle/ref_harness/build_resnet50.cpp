// ORACLE — test infrastructure only.  The reference's ResNet-50 program (resnet/resnet50.cpp: createEngine :154-229, APIToModel :231-246)
// compiled unmodified; ref_build_resnet50 runs ITS APIToModel from run_dir (where "../resnet50.wts" must resolve).
#define main ref_main_resnet50
#include "resnet50.cpp"
#undef main
#include "build_include_main.h"

REF_EXPORT int ref_build_resnet50(const char* run_dir, int max_batch, void** out, size_t* len) {
    if (chdir(run_dir) != 0) return 2;
    nvinfer1::IHostMemory* m = nullptr;
    APIToModel((unsigned)max_batch, &m);
    return ref_copy_out(m, out, len);
}
