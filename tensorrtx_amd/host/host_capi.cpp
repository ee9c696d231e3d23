// extern "C" doorway so tests/bench (ctypes) can drive the C++ host builders.  `options` is a flat
// "key=value;key=value" string.  Returns a malloc'd copy of the serialized plan.
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>

#include "calibrator.h"
#include "common.h"
#include "models.h"
#include "options.h"

using namespace nvinfer1;

static std::map<std::string, std::string> parse_opts(const char* s) {
    std::map<std::string, std::string> m;
    std::stringstream ss(s ? s : "");
    std::string kv;
    while (std::getline(ss, kv, ';')) {
        const auto eq = kv.find('=');
        if (eq != std::string::npos) m[kv.substr(0, eq)] = kv.substr(eq + 1);
    }
    return m;
}
static int geti(const std::map<std::string, std::string>& m, const char* k, int d) {
    auto it = m.find(k);
    return it == m.end() ? d : std::atoi(it->second.c_str());
}

// calibrator used by builds with int8=1: a C v-table supplied by the caller (tests / Python tools), or, when none is set,
// the reference-style directory calibrator over the calibration directory (*.ppm) and cache file of host/options.h (calib_dir= / calib_table= or the environment)
static trtx_calibrator_vtbl g_calib{};
static bool g_have_calib = false;
extern "C" void trtx_host_set_calibrator(const trtx_calibrator_vtbl* v) {
    g_have_calib = v != nullptr;
    if (v) g_calib = *v;
}

extern "C" int32_t trtx_host_build(const char* model, const char* wts_path, const char* options, void** blob, size_t* size) {
    if (!model || !wts_path || !blob || !size) return TRTX_ERR_INVALID;
    const auto o = parse_opts(options);
    trtx_host::Logger logger;
    std::unique_ptr<IBuilder> builder(createInferBuilder(logger));
    std::unique_ptr<IBuilderConfig> config(builder->createBuilderConfig());
    std::unique_ptr<IInt8Calibrator> calibrator;
    if (geti(o, "int8", 0)) {  // USE_INT8 of the reference builders (yolov8/src/model.cpp:317-324, retinaface/retina_r50.cpp:219-225)
        if (!builder->platformHasFastInt8()) return TRTX_ERR_UNSUPPORTED;
        config->setFlag(BuilderFlag::kINT8);
        if (g_have_calib) {
            calibrator.reset(new trtx_host::CallbackCalibrator(g_calib));
        } else {
            const trtx_host::HostOptions ho = trtx_host::read_host_options(o);   // calibration directory / cache: option string, else environment (host/options.h)
            const std::string m0(model);
            calibrator.reset(new trtx_host::Int8EntropyCalibrator2(geti(o, "calib_batch", 1), geti(o, "w", 640), geti(o, "h", 640), ho.calib_dir.c_str(),
                                                                   ho.calib_table.c_str(), m0 == "yolov8n" ? "images" : "data"));
        }
        config->setInt8Calibrator(calibrator.get());
    }
    if (o.count("aux_streams")) {
        const int aux = geti(o, "aux_streams", -1);
        if (aux < -1 || aux > 15) return TRTX_ERR_INVALID;  // the shim's setter is void (as TensorRT's): range-check here
        config->setMaxAuxStreams(aux);
    }
    std::unique_ptr<IHostMemory> plan;
    const std::string m(model);
    if (m == "lenet") {
        plan.reset(trtx_host::buildLenet(builder.get(), config.get(), wts_path, geti(o, "batch", 1)));
    } else if (m == "resnet50") {
        plan.reset(trtx_host::buildResnet50(builder.get(), config.get(), wts_path, geti(o, "batch", 1), geti(o, "fp16", 1) != 0,
                                            geti(o, "h", 224), geti(o, "w", 224)));
    } else if (m == "retinaface_r50") {
        plan.reset(trtx_host::buildRetinaFaceR50(builder.get(), config.get(), wts_path, geti(o, "batch", 1), geti(o, "fp16", 1) != 0,
                                                 geti(o, "h", 480), geti(o, "w", 640)));
    } else if (m == "yolov8n") {
        trtx_host::Yolov8Config cfg;
        cfg.max_batch = geti(o, "batch", 1);
        cfg.fp16 = geti(o, "fp16", 1) != 0;
        cfg.input_h = geti(o, "h", 640);
        cfg.input_w = geti(o, "w", 640);
        cfg.num_class = geti(o, "classes", 80);
        cfg.max_out_bbox = geti(o, "max_out", 1000);
        cfg.mark_heads = geti(o, "mark_heads", 0) != 0;
        cfg.task = geti(o, "task", 0);  // 0 det, 1 seg, 2 pose, 3 obb
        cfg.num_points = geti(o, "points", cfg.num_points);
        if (cfg.task < 0 || cfg.task > 3) return TRTX_ERR_INVALID;
        plan.reset(trtx_host::buildEngineYolov8Det(builder.get(), config.get(), wts_path, cfg));
    } else if (m == "rcnn_r50c4") {
        trtx_host::RcnnConfig cfg;
        cfg.max_batch = geti(o, "batch", 1);
        cfg.fp16 = geti(o, "fp16", 1) != 0;
        cfg.input_h = geti(o, "h", cfg.input_h);
        cfg.input_w = geti(o, "w", cfg.input_w);
        cfg.num_classes = geti(o, "classes", cfg.num_classes);
        cfg.pre_nms_topk = geti(o, "pre_nms_topk", cfg.pre_nms_topk);
        cfg.post_nms_topk = geti(o, "post_nms_topk", cfg.post_nms_topk);
        cfg.detections_per_image = geti(o, "detections", cfg.detections_per_image);
        cfg.nms_method = geti(o, "nms_method", cfg.nms_method);
        cfg.mark_stages = geti(o, "mark_stages", 0) != 0;
        cfg.mask_on = geti(o, "mask", 0) != 0;
        plan.reset(trtx_host::buildRcnnR50C4(builder.get(), config.get(), wts_path, cfg));
    } else {
        return TRTX_ERR_INVALID;
    }
    if (!plan) return TRTX_ERR_UNSUPPORTED;
    *size = plan->size();
    *blob = std::malloc(*size);
    std::memcpy(*blob, plan->data(), *size);
    return TRTX_OK;
}

extern "C" void trtx_host_free(void* blob) { std::free(blob); }
