#include "options.h"

#include <stdlib.h>

namespace trtx {
namespace {
const char* env(const char* name) { return getenv(name); }   // the library's only getenv
int env_int(const char* name, int unset) {
    const char* v = env(name);
    return (v && *v) ? atoi(v) : unset;
}
bool env_is(const char* name, int value) {
    const char* v = env(name);
    return v && *v && atoi(v) == value;
}
bool env_set(const char* name) { return env(name) != nullptr; }
}  // namespace

Options read_options() {
    Options o;
    o.tune = env_int("TRTX_TUNE", -1);
    o.tune_verbose = env_set("TRTX_TUNE_VERBOSE");
    o.tune_margin = env_int("TRTX_TUNE_MARGIN", -1);
    if (const char* v = env("TRTX_TACTIC_CACHE")) o.tactic_cache = v;
    o.graph = env_is("TRTX_GRAPH", 1);
    if (const char* v = env("TRTX_CALIB_REPORT")) o.calib_report = v;
    if (const char* v = env("TRTX_INT8_CLIP_LIMIT"); v && *v) o.int8_clip_limit = atof(v);
    o.lanes = env_int("TRTX_LANES", 0);
    o.group_convs = !env_is("TRTX_GROUP_CONVS", 0);
    o.fold_upsample = !env_is("TRTX_FOLD_UPSAMPLE", 0);
    o.ws = !env_set("TRTX_CONV_NOWS");
    o.wsk = !env_set("TRTX_CONV_NOWSK");
    o.gemm256 = !env_is("TRTX_GEMM256", 0);
    o.patch = !env_is("TRTX_CONV_PATCH", 0);
    o.res = env_int("TRTX_CONV_RES", 7);
    o.roles = !env_is("TRTX_CONV_ROLES", 0);
    o.f32_mfma = !env_set("TRTX_F32_DIRECT");
    o.roialign_fused = !env_set("TRTX_ROIALIGN_PLUGIN");
    o.roialign_fold_stride = !env_is("TRTX_ROIALIGN_FOLD_STRIDE", 0);
    o.profile_kernel_events = !env_set("TRTX_PROFILE_NO_KERNEL_EVENTS");
    o.op_reps = env_int("TRTX_OP_REPS", 1);
    o.conv_dbg = env_int("TRTX_CONV_DBG", 0);
    o.f32_stages = env_int("TRTX_F32_NST", 0);
    return o;
}

const Options& options() {
    static const Options o = read_options();
    return o;
}

}  // namespace trtx
