// Shared host-side helpers for the model builders: .wts loading into nvinfer1::Weights maps
// (the reference copies loadWeights into every model directory: lenet/utils.h:49-80,
// yolov8/src/block.cpp:13-43, rcnn/common.hpp:24-55) and a minimal ILogger (lenet/logging.h:186-220).
#pragma once
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "NvInfer.h"

namespace trtx_host {

using WeightMap = std::map<std::string, nvinfer1::Weights>;

class Logger : public nvinfer1::ILogger {
   public:
    explicit Logger(Severity s = Severity::kWARNING) : mLevel(s) {}
    void log(Severity severity, const char* msg) noexcept override {
        if (severity <= mLevel) std::cerr << "[trtx] " << msg << std::endl;
    }

   private:
    Severity mLevel;
};

// Parses the file through the runtime's loader (trtx_wts_load) and hands back caller-owned malloc'd
// blobs, as the reference does: the builder frees every blob once the engine is built.
inline WeightMap loadWeights(const std::string& file) {
    WeightMap m;
    trtx_wts* w = nullptr;
    const int32_t st = trtx_wts_load(file.c_str(), &w);
    if (st != TRTX_OK) {
        std::cerr << "Unable to load weight file " << file << " (" << trtx_status_string(st) << ")" << std::endl;
        std::abort();  // reference: assert(input.is_open())
    }
    const int32_t n = trtx_wts_count(w);
    for (int32_t i = 0; i < n; ++i) {
        const char* name;
        const float* vals;
        int64_t count;
        trtx_wts_entry(w, i, &name, &vals, &count);
        float* copy = static_cast<float*>(std::malloc(sizeof(float) * (count > 0 ? count : 1)));
        std::memcpy(copy, vals, sizeof(float) * count);
        m[name] = nvinfer1::Weights{nvinfer1::DataType::kFLOAT, copy, count};
    }
    trtx_wts_free(w);
    return m;
}

inline void freeWeights(WeightMap& m) {
    for (auto& kv : m) std::free(const_cast<void*>(kv.second.values));
    m.clear();
}

inline const nvinfer1::Weights& need(const WeightMap& m, const std::string& key) {
    auto it = m.find(key);
    if (it == m.end()) {
        std::cerr << "weight '" << key << "' missing from the .wts file" << std::endl;
        std::abort();
    }
    return it->second;
}

inline nvinfer1::Weights noWeights() { return nvinfer1::Weights{nvinfer1::DataType::kFLOAT, nullptr, 0}; }

// BatchNorm as an IScaleLayer, folded on the host exactly like the reference's addBatchNorm2d
// (yolov8/src/block.cpp:45-77, resnet/resnet50.cpp:77-109): scale = g/sqrt(var+eps), shift = b - mean*scale.
// The temporary blobs are stashed in the map so the final freeWeights releases them.
//
// `sqrt_in_double`: the reference writes an UNQUALIFIED sqrt(var[i] + eps) on floats.  yolov8/src/block.cpp includes <math.h> (:3), where
// libstdc++ brings the float overload into the global namespace, so its fold is float arithmetic throughout.  resnet/resnet50.cpp (:10)
// and retinaface/common.hpp include only <cmath>, which leaves just the C library's ::sqrt(double) visible there: the sum var+eps is
// formed in float, but the root, the division and (for the shift) the subtraction run in double and round to float once, on the store.
// The two differ in the last place of ~60 % of the folded values; the reference's own sources compiled against this shim
// (oracle/ref_build.py, tests/test_ref_builders.py) are what pins which one each model takes.
inline nvinfer1::IScaleLayer* addBatchNorm2d(nvinfer1::INetworkDefinition* network, WeightMap& m, nvinfer1::ITensor& input,
                                             const std::string& lname, float eps, bool sqrt_in_double = false) {
    const float* gamma = static_cast<const float*>(need(m, lname + ".weight").values);
    const float* beta = static_cast<const float*>(need(m, lname + ".bias").values);
    const float* mean = static_cast<const float*>(need(m, lname + ".running_mean").values);
    const float* var = static_cast<const float*>(need(m, lname + ".running_var").values);
    const int64_t len = need(m, lname + ".running_var").count;
    float* sc = static_cast<float*>(std::malloc(sizeof(float) * len));
    float* sh = static_cast<float*>(std::malloc(sizeof(float) * len));
    float* pw = static_cast<float*>(std::malloc(sizeof(float) * len));
    for (int64_t i = 0; i < len; ++i) {
        if (sqrt_in_double) {
            const double root = std::sqrt(static_cast<double>(var[i] + eps));
            sc[i] = static_cast<float>(gamma[i] / root);
            sh[i] = static_cast<float>(beta[i] - mean[i] * gamma[i] / root);
        } else {
            sc[i] = gamma[i] / std::sqrt(var[i] + eps);
            sh[i] = beta[i] - mean[i] * gamma[i] / std::sqrt(var[i] + eps);
        }
        pw[i] = 1.0f;
    }
    using nvinfer1::DataType;
    using nvinfer1::Weights;
    m[lname + ".scale"] = Weights{DataType::kFLOAT, sc, len};
    m[lname + ".shift"] = Weights{DataType::kFLOAT, sh, len};
    m[lname + ".power"] = Weights{DataType::kFLOAT, pw, len};
    auto* l = network->addScale(input, nvinfer1::ScaleMode::kCHANNEL, m[lname + ".shift"], m[lname + ".scale"], m[lname + ".power"]);
    assert(l);
    return l;
}

}  // namespace trtx_host
