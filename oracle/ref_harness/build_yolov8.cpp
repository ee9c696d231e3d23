// ORACLE — test infrastructure only.  Harness around the REFERENCE'S OWN YOLOv8 network builder (yolov8/src/model.cpp:98-336 +
// block.cpp, compiled unmodified where they lie under /root/reference by oracle/ref_build.py): runs buildEngineYolov8Det on a .wts file
// through this repository's nvinfer1 shim and hands back the serialized plan, so that tests can require the product's own host builder
// (tensorrtx_amd/host/yolov8.cpp) to emit the same network, layer for layer and weight for weight.
#include <cstdlib>
#include <cstring>
#include <string>

#include "NvInfer.h"
#include "model.h"  // the reference's header (yolov8/include/model.h)

namespace {
class QuietLogger : public nvinfer1::ILogger {
    void log(Severity, const char*) noexcept override {}
} g_logger;
}  // namespace

// task: 0 detect (buildEngineYolov8Det :98), 1 segment (buildEngineYolov8Seg :1057), 2 pose (buildEngineYolov8Pose :1310), 3 oriented boxes
// (buildEngineYolov8Obb :2499) -- the same numbering tensorrtx_amd/host/host_capi.cpp uses for its `task` option.
extern "C" __attribute__((visibility("default"))) int ref_build_yolov8(int task, const char* wts_path, float gd, float gw, int max_channels, int fp16, void** out,
                                                                       size_t* len) {
    nvinfer1::IBuilder* builder = nvinfer1::createInferBuilder(g_logger);
    nvinfer1::IBuilderConfig* config = builder->createBuilderConfig();
    if (fp16) config->setFlag(nvinfer1::BuilderFlag::kFP16);  // what the reference's serialize_engine does under USE_FP16 (yolov8_det.cpp:27-36 -> model.cpp:313-316)
    const nvinfer1::DataType dt = nvinfer1::DataType::kFLOAT;
    nvinfer1::IHostMemory* m = nullptr;
    switch (task) {
        case 0: m = buildEngineYolov8Det(builder, config, dt, wts_path, gd, gw, max_channels); break;
        case 1: m = buildEngineYolov8Seg(builder, config, dt, wts_path, gd, gw, max_channels); break;
        case 2: m = buildEngineYolov8Pose(builder, config, dt, wts_path, gd, gw, max_channels); break;
        case 3: m = buildEngineYolov8Obb(builder, config, dt, wts_path, gd, gw, max_channels); break;
        default: return 2;
    }
    if (!m) return 1;
    *len = m->size();
    *out = malloc(m->size());
    memcpy(*out, m->data(), m->size());
    delete m;
    delete config;
    delete builder;
    return 0;
}
extern "C" __attribute__((visibility("default"))) int ref_build_yolov8_det(const char* wts_path, float gd, float gw, int max_channels, int fp16, void** out, size_t* len) {
    return ref_build_yolov8(0, wts_path, gd, gw, max_channels, fp16, out, len);
}
extern "C" __attribute__((visibility("default"))) void ref_build_free(void* p) { free(p); }
