// ORACLE — test infrastructure only.  The reference's LeNet-5 program (lenet/lenet.cpp: createLenetEngine :52-156, APIToModel :164-185)
// compiled unmodified; ref_build_lenet runs ITS APIToModel from run_dir (where its WTS_PATH "../models/lenet.wts" must resolve).  The
// reference builds an ICudaEngine and serializes it; without a GPU the shim's engine is the lazy one (include/NvInfer.h), whose
// serialize() returns the plan buildSerializedNetwork produced.
#define main ref_main_lenet
#include "lenet.cpp"
#undef main
#include "build_include_main.h"

REF_EXPORT int ref_build_lenet(const char* run_dir, int max_batch, void** out, size_t* len) {
    if (chdir(run_dir) != 0) return 2;
    nvinfer1::IHostMemory* m = nullptr;
    nvinfer1::IRuntime* runtime = nvinfer1::createInferRuntime(gLogger);
    APIToModel(max_batch, runtime, &m);
    return ref_copy_out(m, out, len);
}
