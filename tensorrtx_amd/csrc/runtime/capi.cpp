// C ABI section 2, builder side: IBuilder / IBuilderConfig / INetworkDefinition (include/trtx_hip.h).
#include <string.h>

#include <memory>

#include "../common.h"
#include "../options.h"
#include "engine.h"
#include "graph.h"
#include "int8.h"
#include "plan.h"
#include "plugin.h"

using namespace trtx;

struct trtx_builder {
    int max_batch = 1;
    bool fp16 = false;
    bool int8 = false;
    int max_aux_streams = -1;
    size_t workspace = 0;
    trtx_calibrator_vtbl calib{};  // IBuilderConfig::setInt8Calibrator
};

struct trtx_network {
    Network net;
    trtx_builder* builder;
    explicit trtx_network(uint32_t flags, trtx_builder* b) : net(flags), builder(b) {}
};

struct trtx_hostmem;
trtx_hostmem* trtx_hostmem_from(std::vector<uint8_t>&& v);

namespace {
std::vector<float> vec(const float* p, int64_t n) {
    if (!p || n <= 0) return {};
    return std::vector<float>(p, p + n);
}
bool valid_layer(const trtx_network* n, int32_t l) { return n && l >= 0 && l < (int32_t)n->net.layers.size(); }
bool valid_tensor(const trtx_network* n, int32_t t) { return n && t >= 0 && t < (int32_t)n->net.tensors.size(); }
}  // namespace

extern "C" int32_t trtx_builder_create(trtx_builder** out) {
    if (!out) return TRTX_ERR_INVALID;
    *out = new trtx_builder();
    return TRTX_OK;
}
extern "C" void trtx_builder_destroy(trtx_builder* b) { delete b; }
extern "C" int32_t trtx_builder_set_max_batch(trtx_builder* b, int32_t n) {
    if (!b || n < 1) return TRTX_ERR_INVALID;
    b->max_batch = n;
    return TRTX_OK;
}
extern "C" int32_t trtx_builder_set_flag(trtx_builder* b, int32_t flag, int32_t on) {
    if (!b) return TRTX_ERR_INVALID;
    if (flag == TRTX_FLAG_FP16)
        b->fp16 = on != 0;
    else if (flag == TRTX_FLAG_INT8)
        b->int8 = on != 0;
    else
        return TRTX_ERR_INVALID;
    return TRTX_OK;
}
extern "C" int32_t trtx_builder_set_max_aux_streams(trtx_builder* b, int32_t n) {
    if (!b || n < -1 || n > 15) return TRTX_ERR_INVALID;
    b->max_aux_streams = n;
    return TRTX_OK;
}
extern "C" int32_t trtx_builder_set_int8_calibrator(trtx_builder* b, const trtx_calibrator_vtbl* calibrator) {
    if (!b) return TRTX_ERR_INVALID;
    b->calib = calibrator ? *calibrator : trtx_calibrator_vtbl{};
    return TRTX_OK;
}
extern "C" int32_t trtx_builder_set_workspace(trtx_builder* b, size_t bytes) {
    if (!b) return TRTX_ERR_INVALID;
    b->workspace = bytes;
    return TRTX_OK;
}

extern "C" int32_t trtx_network_create(trtx_builder* b, uint32_t flags, trtx_network** out) {
    if (!b || !out) return TRTX_ERR_INVALID;
    *out = new trtx_network(flags, b);
    return TRTX_OK;
}
extern "C" void trtx_network_destroy(trtx_network* n) { delete n; }
extern "C" const char* trtx_network_last_error(const trtx_network* n) { return n ? n->net.error.c_str() : ""; }

extern "C" int32_t trtx_add_input(trtx_network* n, const char* name, int32_t dtype, const trtx_dims* dims) {
    if (!n || !dims || dims->nb < 1 || dims->nb > 8) return -1;
    return n->net.add_input(name, dtype, from_c(*dims));
}

static int32_t add_conv_like(trtx_network* n, int kind, int32_t input, int32_t nb_out, int32_t kh, int32_t kw,
                             const float* kernel, int64_t kc, const float* bias, int64_t bc) {
    if (!valid_tensor(n, input) || nb_out < 1 || kh < 1 || kw < 1) return -1;
    LayerDef l;
    l.kind = kind;
    l.inputs = {input};
    l.nb_out = nb_out;
    l.kernel[0] = kh;
    l.kernel[1] = kw;
    l.w0 = vec(kernel, kc);
    l.w1 = vec(bias, bc);
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_convolution(trtx_network* n, int32_t input, int32_t nb_out, int32_t kh, int32_t kw,
                                        const float* kernel, int64_t kc, const float* bias, int64_t bc) {
    return add_conv_like(n, L_CONV, input, nb_out, kh, kw, kernel, kc, bias, bc);
}
extern "C" int32_t trtx_add_deconvolution(trtx_network* n, int32_t input, int32_t nb_out, int32_t kh, int32_t kw,
                                          const float* kernel, int64_t kc, const float* bias, int64_t bc) {
    return add_conv_like(n, L_DECONV, input, nb_out, kh, kw, kernel, kc, bias, bc);
}
extern "C" int32_t trtx_add_fully_connected(trtx_network* n, int32_t input, int32_t nb_out, const float* kernel,
                                            int64_t kc, const float* bias, int64_t bc) {
    return add_conv_like(n, L_FULLY_CONNECTED, input, nb_out, 1, 1, kernel, kc, bias, bc);
}
extern "C" int32_t trtx_add_activation(trtx_network* n, int32_t input, int32_t type) {
    if (!valid_tensor(n, input)) return -1;
    LayerDef l;
    l.kind = L_ACTIVATION;
    l.inputs = {input};
    l.op = type;
    if (type == TRTX_ACTIVATION_LEAKY_RELU) l.alpha = 0.01f;
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_pooling(trtx_network* n, int32_t input, int32_t type, int32_t kh, int32_t kw) {
    if (!valid_tensor(n, input) || kh < 1 || kw < 1) return -1;
    LayerDef l;
    l.kind = L_POOLING;
    l.inputs = {input};
    l.op = type;
    l.kernel[0] = kh;
    l.kernel[1] = kw;
    // TensorRT's default pooling stride is the window size... no: it is 1; the reference always sets it explicitly
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_scale(trtx_network* n, int32_t input, int32_t mode, const float* shift, int64_t shc,
                                  const float* scale, int64_t scc, const float* power, int64_t pc) {
    if (!valid_tensor(n, input)) return -1;
    LayerDef l;
    l.kind = L_SCALE;
    l.inputs = {input};
    l.op = mode;
    l.w0 = vec(shift, shc);
    l.w1 = vec(scale, scc);
    l.w2 = vec(power, pc);
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_elementwise(trtx_network* n, int32_t a, int32_t b, int32_t op) {
    if (!valid_tensor(n, a) || !valid_tensor(n, b)) return -1;
    LayerDef l;
    l.kind = L_ELEMENTWISE;
    l.inputs = {a, b};
    l.op = op;
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_concatenation(trtx_network* n, const int32_t* inputs, int32_t nb) {
    if (!n || !inputs || nb < 1) return -1;
    LayerDef l;
    l.kind = L_CONCAT;
    for (int i = 0; i < nb; ++i) {
        if (!valid_tensor(n, inputs[i])) return -1;
        l.inputs.push_back(inputs[i]);
    }
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_slice(trtx_network* n, int32_t input, const trtx_dims* start, const trtx_dims* size,
                                  const trtx_dims* stride) {
    if (!valid_tensor(n, input) || !start || !size || !stride) return -1;
    LayerDef l;
    l.kind = L_SLICE;
    l.inputs = {input};
    l.start = from_c(*start);
    l.size = from_c(*size);
    l.step = from_c(*stride);
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_shuffle(trtx_network* n, int32_t input) {
    if (!valid_tensor(n, input)) return -1;
    LayerDef l;
    l.kind = L_SHUFFLE;
    l.inputs = {input};
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_resize(trtx_network* n, int32_t input) {
    if (!valid_tensor(n, input)) return -1;
    LayerDef l;
    l.kind = L_RESIZE;
    l.inputs = {input};
    l.op = TRTX_RESIZE_NEAREST;
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_softmax(trtx_network* n, int32_t input) {
    if (!valid_tensor(n, input)) return -1;
    LayerDef l;
    l.kind = L_SOFTMAX;
    l.inputs = {input};
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_matrix_multiply(trtx_network* n, int32_t a, int32_t op_a, int32_t b, int32_t op_b) {
    if (!valid_tensor(n, a) || !valid_tensor(n, b)) return -1;
    LayerDef l;
    l.kind = L_MATMUL;
    l.inputs = {a, b};
    l.mm_op[0] = op_a;
    l.mm_op[1] = op_b;
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_constant(trtx_network* n, const trtx_dims* dims, const float* values, int64_t count) {
    if (!n || !dims) return -1;
    LayerDef l;
    l.kind = L_CONSTANT;
    l.out_dims = from_c(*dims);
    l.w0 = vec(values, count);
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_reduce(trtx_network* n, int32_t input, int32_t op, uint32_t axes, int32_t keep_dims) {
    if (!valid_tensor(n, input)) return -1;
    LayerDef l;
    l.kind = L_REDUCE;
    l.inputs = {input};
    l.op = op;
    l.axis = (int32_t)axes;
    l.keep_dims = keep_dims;
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_identity(trtx_network* n, int32_t input) {
    if (!valid_tensor(n, input)) return -1;
    LayerDef l;
    l.kind = L_IDENTITY;
    l.inputs = {input};
    return n->net.add_layer(std::move(l));
}
extern "C" int32_t trtx_add_plugin_v2(trtx_network* n, const int32_t* inputs, int32_t nb, const trtx_plugin_vtbl* plugin) {
    if (!n || !inputs || nb < 1 || !plugin || !plugin->clone) return -1;
    trtx_plugin_vtbl copy{};
    if (plugin->clone(plugin->self, &copy) != 0) {  // TensorRT clones the plugin handed to addPluginV2
        n->net.error = "plugin clone() failed";
        return -1;
    }
    LayerDef l;
    l.kind = L_PLUGIN;
    for (int i = 0; i < nb; ++i) {
        if (!valid_tensor(n, inputs[i])) return -1;
        l.inputs.push_back(inputs[i]);
    }
    l.plugin = std::make_shared<PluginHolder>(copy);
    return n->net.add_layer(std::move(l));
}

extern "C" int32_t trtx_layer_nb_outputs(const trtx_network* n, int32_t layer) {
    return valid_layer(n, layer) ? (int32_t)n->net.layers[layer].outputs.size() : 0;
}
extern "C" int32_t trtx_layer_output(const trtx_network* n, int32_t layer, int32_t index) {
    if (!valid_layer(n, layer) || index < 0 || index >= (int32_t)n->net.layers[layer].outputs.size()) return -1;
    return n->net.layers[layer].outputs[index];
}
extern "C" int32_t trtx_layer_set_name(trtx_network* n, int32_t layer, const char* name) {
    if (!valid_layer(n, layer) || !name) return TRTX_ERR_INVALID;
    n->net.layers[layer].name = name;
    return TRTX_OK;
}

static int32_t reinfer(trtx_network* n, int32_t layer) {
    // like TensorRT, a setter never fails on a transiently inconsistent layer; build-time validation does
    if (n->net.infer(layer))
        n->net.error.clear();
    else
        for (int t : n->net.layers[layer].outputs) n->net.tensors[t].dims = Dims{};
    return TRTX_OK;
}

extern "C" int32_t trtx_layer_set_ints(trtx_network* n, int32_t layer, int32_t param, const int32_t* v, int32_t count) {
    if (!valid_layer(n, layer) || !v || count < 1) return TRTX_ERR_INVALID;
    LayerDef& l = n->net.layers[layer];
    auto two = [&](int32_t* dst) {
        dst[0] = v[0];
        dst[1] = count > 1 ? v[1] : v[0];
    };
    // reject values that shape inference would divide by (SIGFPE inside the C ABI otherwise)
    if ((param == TRTX_P_STRIDE || param == TRTX_P_DILATION || param == TRTX_P_KERNEL) && (v[0] < 1 || (count > 1 && v[1] < 1))) return TRTX_ERR_INVALID;
    if (param == TRTX_P_GROUPS && v[0] < 1) return TRTX_ERR_INVALID;
    if (param == TRTX_P_PADDING && (v[0] < 0 || (count > 1 && v[1] < 0))) return TRTX_ERR_INVALID;
    switch (param) {
        case TRTX_P_STRIDE: two(l.stride); break;
        case TRTX_P_PADDING: two(l.padding); break;
        case TRTX_P_DILATION: two(l.dilation); break;
        case TRTX_P_KERNEL: two(l.kernel); break;
        case TRTX_P_GROUPS: l.groups = v[0]; break;
        case TRTX_P_NB_OUT: l.nb_out = v[0]; break;
        case TRTX_P_AXIS: l.axis = v[0]; break;
        case TRTX_P_AVG_EXCLUSIVE: l.avg_exclusive = v[0]; break;
        case TRTX_P_RESIZE_MODE: l.op = v[0]; break;
        case TRTX_P_FIRST_TRANSPOSE:
        case TRTX_P_SECOND_TRANSPOSE: {
            int32_t* dst = param == TRTX_P_FIRST_TRANSPOSE ? l.perm1 : l.perm2;
            for (int i = 0; i < 8; ++i) dst[i] = i < count ? v[i] : i;
            break;
        }
        default: return TRTX_ERR_INVALID;
    }
    return reinfer(n, layer);
}
extern "C" int32_t trtx_layer_set_floats(trtx_network* n, int32_t layer, int32_t param, const float* v, int32_t count) {
    if (!valid_layer(n, layer) || !v || count < 1) return TRTX_ERR_INVALID;
    LayerDef& l = n->net.layers[layer];
    switch (param) {
        case TRTX_P_ALPHA: l.alpha = v[0]; break;
        case TRTX_P_BETA: l.beta = v[0]; break;
        case TRTX_P_RESIZE_SCALES:
            if (count > 8) return TRTX_ERR_INVALID;
            l.nb_scales = count;
            for (int i = 0; i < count; ++i) l.scales[i] = v[i];
            l.out_dims = Dims{};
            break;
        default: return TRTX_ERR_INVALID;
    }
    return reinfer(n, layer);
}
extern "C" int32_t trtx_layer_set_dims(trtx_network* n, int32_t layer, int32_t param, const trtx_dims* d) {
    if (!valid_layer(n, layer) || !d) return TRTX_ERR_INVALID;
    LayerDef& l = n->net.layers[layer];
    switch (param) {
        case TRTX_P_RESHAPE: l.reshape = from_c(*d); break;
        case TRTX_P_RESIZE_OUT_DIMS:
            l.out_dims = from_c(*d);
            l.nb_scales = 0;
            break;
        default: return TRTX_ERR_INVALID;
    }
    return reinfer(n, layer);
}
extern "C" int32_t trtx_tensor_get_dims(const trtx_network* n, int32_t tensor, trtx_dims* out) {
    if (!valid_tensor(n, tensor) || !out) return TRTX_ERR_INVALID;
    *out = to_c(n->net.tensors[tensor].dims);
    return TRTX_OK;
}
extern "C" int32_t trtx_tensor_set_name(trtx_network* n, int32_t tensor, const char* name) {
    if (!valid_tensor(n, tensor) || !name) return TRTX_ERR_INVALID;
    n->net.tensors[tensor].name = name;
    return TRTX_OK;
}
extern "C" const char* trtx_tensor_get_name(const trtx_network* n, int32_t tensor) {
    return valid_tensor(n, tensor) ? n->net.tensors[tensor].name.c_str() : nullptr;
}
extern "C" int32_t trtx_mark_output(trtx_network* n, int32_t tensor) {
    if (!valid_tensor(n, tensor)) return TRTX_ERR_INVALID;
    return n->net.mark_output(tensor) ? TRTX_OK : TRTX_ERR_INVALID;
}

extern "C" int32_t trtx_build_serialized(trtx_builder* b, trtx_network* n, trtx_hostmem** out) {
    if (!b || !n || !out) return TRTX_ERR_INVALID;
    n->net.max_batch = b->max_batch;
    n->net.fp16 = b->fp16;
    n->net.max_aux_streams = b->max_aux_streams;
    n->net.int8 = false;
    n->net.tensor_scale.clear();
    if (n->net.output_ids().empty()) {
        n->net.error = "network has no outputs";
        return TRTX_ERR_STATE;
    }
    if (!n->net.validate()) {
        fprintf(stderr, "[trtx_hip] buildSerializedNetwork: %s\n", n->net.error.c_str());
        return TRTX_ERR_INVALID;
    }
    if (b->int8) {
        // kINT8 (yolov8/src/model.cpp:317-324): activation scales from the calibrator's cache, or from a calibration run on the GPU.
        // Layers that cannot run in int8 fall back to fp16 (the engine is built as fp16 + int8).
        n->net.fp16 = true;
        bool have = false;
        if (b->calib.read_cache) {
            size_t len = 0;
            const void* cache = b->calib.read_cache(b->calib.self, &len);
            if (cache && len) {
                std::string err;
                if (!read_calibration_cache(cache, len, &n->net, &err)) {
                    fprintf(stderr, "[trtx_hip] buildSerializedNetwork: %s\n", err.c_str());
                    return TRTX_ERR_IO;
                }
                have = true;
            }
        }
        if (!have) {
            if (!b->calib.get_batch) {
                fprintf(stderr, "[trtx_hip] buildSerializedNetwork: BuilderFlag::kINT8 needs setInt8Calibrator (batches or a calibration cache)\n");
                return TRTX_ERR_STATE;
            }
            const int32_t st = run_int8_calibration(&n->net, b->calib);
            if (st != TRTX_OK) return st;
            if (b->calib.write_cache) {
                const std::string text = write_calibration_cache(n->net);
                b->calib.write_cache(b->calib.self, text.data(), text.size());
            }
        }
        n->net.int8 = true;
    }
    // validate by lowering once on the host (no device needed): unsupported graphs fail at build time
    Plan plan;
    if (!lower_network(n->net, &plan)) {
        n->net.error = plan.error;
        fprintf(stderr, "[trtx_hip] buildSerializedNetwork: %s\n", plan.error.c_str());
        return TRTX_ERR_UNSUPPORTED;
    }
    std::vector<uint8_t> blob;
    n->net.tactics.clear();
    n->net.tactics_timed = false;
    n->net.serialize(blob);
    // Kernel tactics by timing, where there is a GPU to time on (what TensorRT's builder does at this point): a throw-away engine
    // of this plan times the launch configurations of its convolutions (runtime/tune.cpp) and the choices are stored IN the plan, so
    // that every deserialize of it runs the same kernels.  Without a GPU (or with TRTX_TUNE=0) the plan carries no choices and
    // engines made from it run the static defaults.
    const int tune_opt = read_options().tune;
    // (fp32 engines too since round 5: their convolutions have tile shapes to choose from - kernels/conv_igemm_f32.hip - all of them the same bits)
    if (tune_opt != 0 && trtx_device_count() > 0) {
        trtx_engine* e = nullptr;
        if (engine_from_plan(blob.data(), blob.size(), true, &e) == TRTX_OK && e) {
            n->net.tactics = e->net->tactics;
            n->net.tactics_timed = e->net->tactics_timed;
            trtx_engine_destroy(e);
            blob.clear();
            n->net.serialize(blob);
        } else {
            (void)hipGetLastError();
            fprintf(stderr, "[trtx_hip] buildSerializedNetwork: tactic timing skipped, the plan carries the static defaults\n");
        }
    }
    *out = trtx_hostmem_from(std::move(blob));
    return TRTX_OK;
}

// test hook: the entropy threshold search of the INT8 calibration (int8.cpp) on a caller-supplied |x| histogram
extern "C" float trtx_int8_entropy_threshold(const double* hist, int32_t bins, float range) {
    if (!hist || bins < 128) return 0.f;
    return entropy_threshold(std::vector<double>(hist, hist + bins), range);
}
// ... and the clip limit the calibration applies on top of it (limit <= 0: thr unchanged)
extern "C" float trtx_int8_clip_limited_threshold(const double* hist, int32_t bins, float range, float thr, double limit) {
    if (!hist || bins < 1) return thr;
    return clip_limited_threshold(std::vector<double>(hist, hist + bins), range, thr, limit);
}
