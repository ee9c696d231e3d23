// Runtime-side owner of one plugin instance (driven through the C v-table of include/trtx_hip.h) and the
// process-wide creator registry (getPluginRegistry() of the reference: yolov8/src/block.cpp:263).
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "graph.h"
#include "trtx_hip.h"

namespace trtx {

inline trtx_dims to_c(const Dims& d) {
    trtx_dims o{};
    o.nb = d.nb;
    for (int i = 0; i < 8; ++i) o.d[i] = d.d[i];
    return o;
}
inline Dims from_c(const trtx_dims& d) {
    Dims o;
    o.nb = d.nb < 0 ? 0 : (d.nb > 8 ? 8 : d.nb);
    for (int i = 0; i < 8; ++i) o.d[i] = d.d[i];
    return o;
}

struct PluginHolder {
    trtx_plugin_vtbl v{};
    explicit PluginHolder(const trtx_plugin_vtbl& vt) : v(vt) {}
    ~PluginHolder() {
        if (v.destroy) v.destroy(v.self);
    }
    PluginHolder(const PluginHolder&) = delete;
    PluginHolder& operator=(const PluginHolder&) = delete;

    int nb_outputs() const { return v.get_nb_outputs(v.self); }
    bool output_dims(int idx, const std::vector<Dims>& ins, Dims* out) const {
        std::vector<trtx_dims> ci;
        for (const auto& d : ins) ci.push_back(to_c(d));
        trtx_dims o{};
        if (v.get_output_dims(v.self, idx, ci.data(), (int)ci.size(), &o) != 0) return false;
        *out = from_c(o);
        return true;
    }
    std::string type() const { return v.plugin_type(v.self); }
    // a null version string reads as "1": the reference's MishPlugin::getPluginVersion returns 0 (yolov4/mish.cu:92-95) while its creator
    // registers under "1" - the version a plan must name to find the creator again
    std::string version() const {
        const char* s = v.plugin_version(v.self);
        return s ? s : "1";
    }
    std::vector<uint8_t> serialize() const {
        std::vector<uint8_t> b(v.serialization_size(v.self));
        if (!b.empty()) v.serialize(v.self, b.data());
        return b;
    }
    std::shared_ptr<PluginHolder> clone() const {
        trtx_plugin_vtbl c{};
        if (!v.clone || v.clone(v.self, &c) != 0) return nullptr;
        return std::make_shared<PluginHolder>(c);
    }
};

class PluginRegistry {
   public:
    static PluginRegistry& instance();
    int32_t add(const trtx_creator_vtbl& c);
    bool get(const std::string& name, const std::string& version, trtx_creator_vtbl* out);
    std::shared_ptr<PluginHolder> deserialize(const std::string& type, const std::string& version, const void* data,
                                              size_t len);

   private:
    PluginRegistry();
    std::mutex mu_;
    std::map<std::string, trtx_creator_vtbl> creators_;
};

// built-in HIP plugins (plugins/builtin_plugins.cpp)
void register_builtin_plugins(PluginRegistry& r);
// parameters of a *built-in* YoloLayer_TRT instance (false for any other / user-provided plugin): lets the
// lowering pass replace the DFL + plugin tail by the fused NHWC kernel
struct YoloLayerParams {
    int classes, net_w, net_h, max_out;
    std::vector<int> strides;
    bool det_only;
};
bool builtin_yolo_params(const trtx_plugin_vtbl& v, YoloLayerParams* out);
// true for the built-in "Mish_TRT" (plugins/builtin_plugins.cpp; yolov4/mish.{h,cu}): a pointwise activation, which the lowering pass
// folds into the producing convolution's epilogue (ACT_MISH) or runs as an activation op in the tensor's own layout
bool builtin_is_mish(const trtx_plugin_vtbl& v);
// true for the built-in HIP plugins (kernels + stream-ordered memsets only): their enqueue may run inside a hipGraph capture.
// A user IPluginV2 may synchronise, allocate or copy from pageable memory (the reference's R-CNN plugins do all three).
bool builtin_plugin_capturable(const trtx_plugin_vtbl& v);

}  // namespace trtx
