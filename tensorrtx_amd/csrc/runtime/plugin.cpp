#include "plugin.h"

namespace trtx {

PluginRegistry::PluginRegistry() {}

PluginRegistry& PluginRegistry::instance() {
    static PluginRegistry* r = [] {
        auto* p = new PluginRegistry();
        register_builtin_plugins(*p);
        return p;
    }();
    return *r;
}

int32_t PluginRegistry::add(const trtx_creator_vtbl& c) {
    if (!c.plugin_name || !c.plugin_version || !c.deserialize) return TRTX_ERR_INVALID;
    std::lock_guard<std::mutex> g(mu_);
    const std::string key = std::string(c.plugin_name(c.self)) + "/" + c.plugin_version(c.self);
    creators_[key] = c;  // later registration wins (a user plugin may override a built-in)
    return TRTX_OK;
}

bool PluginRegistry::get(const std::string& name, const std::string& version, trtx_creator_vtbl* out) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = creators_.find(name + "/" + version);
    if (it == creators_.end()) return false;
    if (out) *out = it->second;
    return true;
}

std::shared_ptr<PluginHolder> PluginRegistry::deserialize(const std::string& type, const std::string& version,
                                                          const void* data, size_t len) {
    trtx_creator_vtbl c{};
    if (!get(type, version, &c)) return nullptr;
    trtx_plugin_vtbl v{};
    if (c.deserialize(c.self, type.c_str(), data, len, &v) != 0) return nullptr;
    return std::make_shared<PluginHolder>(v);
}

}  // namespace trtx

extern "C" int32_t trtx_registry_register(const trtx_creator_vtbl* creator) {
    if (!creator) return TRTX_ERR_INVALID;
    return trtx::PluginRegistry::instance().add(*creator);
}

extern "C" int32_t trtx_registry_get(const char* name, const char* version, trtx_creator_vtbl* out) {
    if (!name || !version) return TRTX_ERR_INVALID;
    return trtx::PluginRegistry::instance().get(name, version, out) ? TRTX_OK : TRTX_ERR_INVALID;
}
