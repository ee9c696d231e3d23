// Built-in HIP plugins, registered under the names the reference host code asks the registry for.
//   "YoloLayer_TRT"/"1"  — yolov8/plugin/yololayer.{h,cu} (creator: yololayer.cu:320-369)
// Each plugin is a small host object exposed through the C v-table of include/trtx_hip.h; its
// enqueue calls the HIP operator entry points of section 1.  Serialization blobs keep the reference's
// field order so a plan round-trips byte for byte.
#include <string.h>

#include <vector>

#include "../common.h"
#include "../runtime/plugin.h"

namespace trtx {
namespace {

template <typename T>
void put(std::vector<uint8_t>& b, const T& v) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    b.insert(b.end(), p, p + sizeof(T));
}
template <typename T>
bool get(const uint8_t*& p, const uint8_t* end, T& v) {
    if (p + sizeof(T) > end) return false;
    memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return true;
}

// ------------------------------------------------------------------------------------------------
struct YoloLayer {
    int class_count = 80, n_kpt = 17;
    float kpt_conf = 0.f;
    int thread_count = 256, net_w = 640, net_h = 640, max_out = 1000;
    std::vector<int> strides;
    bool seg = false, pose = false, obb = false;
    int max_batch = 1;

    std::vector<uint8_t> blob() const {  // yololayer.cu:75-101
        std::vector<uint8_t> b;
        put(b, class_count);
        put(b, n_kpt);
        put(b, kpt_conf);
        put(b, thread_count);
        put(b, net_w);
        put(b, net_h);
        put(b, max_out);
        put(b, (int)strides.size());
        for (int s : strides) put(b, s);
        put(b, seg);
        put(b, pose);
        put(b, obb);
        return b;
    }
    static YoloLayer* from_blob(const void* data, size_t len) {  // yololayer.cu:53-73
        const uint8_t* p = static_cast<const uint8_t*>(data);
        const uint8_t* end = p + len;
        auto* y = new YoloLayer();
        int ns = 0;
        bool ok = get(p, end, y->class_count) && get(p, end, y->n_kpt) && get(p, end, y->kpt_conf) &&
                  get(p, end, y->thread_count) && get(p, end, y->net_w) && get(p, end, y->net_h) &&
                  get(p, end, y->max_out) && get(p, end, ns) && ns >= 0 && ns <= 8;
        for (int i = 0; ok && i < ns; ++i) {
            int s = 0;
            ok = get(p, end, s);
            y->strides.push_back(s);
        }
        ok = ok && get(p, end, y->seg) && get(p, end, y->pose) && get(p, end, y->obb) && p == end;
        if (!ok) {
            delete y;
            return nullptr;
        }
        return y;
    }
};

void yolo_fill(trtx_plugin_vtbl* v, YoloLayer* y);

int32_t yolo_nb_outputs(void*) { return 1; }
int32_t yolo_output_dims(void* s, int32_t, const trtx_dims*, int32_t, trtx_dims* out) {
    auto* y = static_cast<YoloLayer*>(s);
    out->nb = 3;  // Dims3(total_size + 1, 1, 1), yololayer.cu:107-111
    out->d[0] = (int64_t)y->max_out * kYoloDetFloats + 1;
    out->d[1] = 1;
    out->d[2] = 1;
    return 0;
}
int32_t yolo_configure(void* s, const trtx_dims* in, int32_t nb_in, const trtx_dims*, int32_t, int32_t max_batch) {
    auto* y = static_cast<YoloLayer*>(s);
    y->max_batch = max_batch;
    if (nb_in != (int)y->strides.size()) return 1;
    for (int i = 0; i < nb_in; ++i) {
        const int64_t cells = (int64_t)(y->net_h / y->strides[i]) * (y->net_w / y->strides[i]);
        int64_t vol = 1;
        for (int k = 0; k < in[i].nb; ++k) vol *= in[i].d[k];
        const int info = 4 + y->class_count + (y->seg ? 32 : 0) + (y->pose ? y->n_kpt * 3 : 0) + (y->obb ? 1 : 0);
        if (vol != cells * info) return 1;  // [4 + classes (+32) (+3*nk) (+1), cells]
    }
    return 0;
}
int32_t yolo_initialize(void* s) {
    auto* y = static_cast<YoloLayer*>(s);
    return (y->n_kpt < 0 || y->n_kpt > 17) ? 1 : 0;  // Detection::keypoints holds kNumberOfPoints = 17 triples
}
void yolo_terminate(void*) {}
size_t yolo_workspace(void* s, int32_t max_batch) {
    auto* y = static_cast<YoloLayer*>(s);
    return trtx_yolo_decode_workspace(max_batch, y->net_h, y->net_w, y->strides.data(), (int)y->strides.size());
}
int32_t yolo_enqueue(void* s, int32_t batch, const void* const* inputs, void* const* outputs, void* ws,
                     trtx_stream_t stream) {
    auto* y = static_cast<YoloLayer*>(s);
    const size_t ws_bytes =
            trtx_yolo_decode_workspace(batch, y->net_h, y->net_w, y->strides.data(), (int)y->strides.size());
    return trtx_yolo_decode_ex(reinterpret_cast<const float* const*>(inputs), (int)y->strides.size(), batch, y->class_count, y->net_h,
                               y->net_w, y->strides.data(), y->max_out, y->n_kpt, y->kpt_conf, y->seg, y->pose, y->obb,
                               static_cast<float*>(outputs[0]), ws, ws_bytes, stream);
}
size_t yolo_ser_size(void* s) { return static_cast<YoloLayer*>(s)->blob().size(); }
void yolo_serialize(void* s, void* buf) {
    const auto b = static_cast<YoloLayer*>(s)->blob();
    memcpy(buf, b.data(), b.size());
}
const char* yolo_type(void*) { return "YoloLayer_TRT"; }
const char* yolo_version(void*) { return "1"; }
int32_t yolo_clone(void* s, trtx_plugin_vtbl* out) {
    yolo_fill(out, new YoloLayer(*static_cast<YoloLayer*>(s)));
    return 0;
}
void yolo_destroy(void* s) { delete static_cast<YoloLayer*>(s); }

void yolo_fill(trtx_plugin_vtbl* v, YoloLayer* y) {
    v->self = y;
    v->get_nb_outputs = yolo_nb_outputs;
    v->get_output_dims = yolo_output_dims;
    v->configure = yolo_configure;
    v->initialize = yolo_initialize;
    v->terminate = yolo_terminate;
    v->workspace_size = yolo_workspace;
    v->enqueue = yolo_enqueue;
    v->serialization_size = yolo_ser_size;
    v->serialize = yolo_serialize;
    v->plugin_type = yolo_type;
    v->plugin_version = yolo_version;
    v->clone = yolo_clone;
    v->destroy = yolo_destroy;
}

// ------------------------------------------------------------------------------------------------
// The anchor-based form of "YoloLayer_TRT"/"1" (yolov5/plugin/yololayer.{h,cu}; same registered name as the YOLOv8 plugin in
// the reference, a different parameter set): creator fields "netinfo" = int32[5] {classes, W, H, maxOut, isSeg} and "kernels" =
// YoloKernel[n] {int width, height; float anchors[6]} (yolov5/src/model.cpp:246-275, yololayer.cu:250-266); blob
// int classCount, threadCount, kernelCount, netW, netH, maxOut, bool isSeg, YoloKernel[n] (yololayer.cu:49-89).
struct Yolo5Kernel {
    int width, height;
    float anchors[6];
};
struct Yolo5Layer {
    int class_count = 80, thread_count = 256, net_w = 640, net_h = 640, max_out = 1000;
    bool seg = false;
    std::vector<Yolo5Kernel> kernels;

    std::vector<uint8_t> blob() const {
        std::vector<uint8_t> b;
        put(b, class_count);
        put(b, thread_count);
        put(b, (int)kernels.size());
        put(b, net_w);
        put(b, net_h);
        put(b, max_out);
        put(b, seg);
        for (const auto& k : kernels) put(b, k);
        return b;
    }
    static Yolo5Layer* from_blob(const void* data, size_t len) {
        const uint8_t* p = static_cast<const uint8_t*>(data);
        const uint8_t* end = p + len;
        auto* y = new Yolo5Layer();
        int nk = 0;
        bool ok = get(p, end, y->class_count) && get(p, end, y->thread_count) && get(p, end, nk) && get(p, end, y->net_w) &&
                  get(p, end, y->net_h) && get(p, end, y->max_out) && get(p, end, y->seg) && nk >= 1 && nk <= 8 &&
                  (size_t)(end - p) == (size_t)nk * sizeof(Yolo5Kernel);
        for (int i = 0; ok && i < nk; ++i) {
            Yolo5Kernel k{};
            ok = get(p, end, k) && k.width > 0 && k.height > 0;
            y->kernels.push_back(k);
        }
        if (!ok || y->class_count < 1 || y->max_out < 1) {
            delete y;
            return nullptr;
        }
        return y;
    }
    void geometry(std::vector<int>* gw, std::vector<int>* gh, std::vector<float>* an) const {
        for (const auto& k : kernels) {
            gw->push_back(k.width);
            gh->push_back(k.height);
            an->insert(an->end(), k.anchors, k.anchors + 6);
        }
    }
};

void yolo5_fill(trtx_plugin_vtbl* v, Yolo5Layer* y);
int32_t yolo5_output_dims(void* s, int32_t, const trtx_dims*, int32_t, trtx_dims* out) {
    auto* y = static_cast<Yolo5Layer*>(s);
    out->nb = 3;  // Dims3(maxOut * sizeof(Detection) / 4 + 1, 1, 1), yololayer.cu:101-105
    out->d[0] = (int64_t)y->max_out * 38 + 1;
    out->d[1] = 1;
    out->d[2] = 1;
    return 0;
}
int32_t yolo5_configure(void* s, const trtx_dims* in, int32_t nb_in, const trtx_dims*, int32_t, int32_t) {
    auto* y = static_cast<Yolo5Layer*>(s);
    if (nb_in != (int)y->kernels.size()) return 1;
    const int info = 5 + y->class_count + (y->seg ? 32 : 0);
    for (int i = 0; i < nb_in; ++i) {
        int64_t vol = 1;
        for (int k = 0; k < in[i].nb; ++k) vol *= in[i].d[k];
        if (vol != (int64_t)3 * info * y->kernels[i].width * y->kernels[i].height) return 1;
    }
    return 0;
}
int32_t yolo5_initialize(void*) { return 0; }
size_t yolo5_workspace(void* s, int32_t max_batch) {
    auto* y = static_cast<Yolo5Layer*>(s);
    std::vector<int> gw, gh;
    std::vector<float> an;
    y->geometry(&gw, &gh, &an);
    return trtx_yolov5_decode_workspace(max_batch, gw.data(), gh.data(), (int)gw.size());
}
int32_t yolo5_enqueue(void* s, int32_t batch, const void* const* inputs, void* const* outputs, void* ws, trtx_stream_t stream) {
    auto* y = static_cast<Yolo5Layer*>(s);
    std::vector<int> gw, gh;
    std::vector<float> an;
    y->geometry(&gw, &gh, &an);
    const size_t ws_bytes = trtx_yolov5_decode_workspace(batch, gw.data(), gh.data(), (int)gw.size());
    return trtx_yolov5_decode(reinterpret_cast<const float* const*>(inputs), (int)gw.size(), batch, y->class_count, y->net_h, y->net_w,
                              gw.data(), gh.data(), an.data(), y->max_out, y->seg ? 1 : 0, static_cast<float*>(outputs[0]), ws, ws_bytes,
                              stream);
}
size_t yolo5_ser_size(void* s) { return static_cast<Yolo5Layer*>(s)->blob().size(); }
void yolo5_serialize(void* s, void* buf) {
    const auto b = static_cast<Yolo5Layer*>(s)->blob();
    memcpy(buf, b.data(), b.size());
}
int32_t yolo5_clone(void* s, trtx_plugin_vtbl* out) {
    yolo5_fill(out, new Yolo5Layer(*static_cast<Yolo5Layer*>(s)));
    return 0;
}
void yolo5_destroy(void* s) { delete static_cast<Yolo5Layer*>(s); }
void yolo5_fill(trtx_plugin_vtbl* v, Yolo5Layer* y) {
    v->self = y;
    v->get_nb_outputs = yolo_nb_outputs;
    v->get_output_dims = yolo5_output_dims;
    v->configure = yolo5_configure;
    v->initialize = yolo5_initialize;
    v->terminate = yolo_terminate;
    v->workspace_size = yolo5_workspace;
    v->enqueue = yolo5_enqueue;
    v->serialization_size = yolo5_ser_size;
    v->serialize = yolo5_serialize;
    v->plugin_type = yolo_type;
    v->plugin_version = yolo_version;
    v->clone = yolo5_clone;
    v->destroy = yolo5_destroy;
}
int32_t yolo5_create(const trtx_plugin_field* f, trtx_plugin_vtbl* out) {
    if (!f[0].data || !f[1].data || f[0].length < 5 || f[1].length < 1 || f[1].length > 8) return 1;
    const int* ni = static_cast<const int*>(f[0].data);
    auto* y = new Yolo5Layer();
    y->class_count = ni[0];
    y->net_w = ni[1];
    y->net_h = ni[2];
    y->max_out = ni[3];
    y->seg = ni[4] != 0;
    y->kernels.resize(f[1].length);  // the reference counts this field in YoloKernel elements (model.cpp:272-273)
    memcpy(y->kernels.data(), f[1].data, sizeof(Yolo5Kernel) * f[1].length);
    yolo5_fill(out, y);
    return 0;
}

// creator: one field "combinedInfo" = int32[9 + nStrides] (yolov8/src/block.cpp:267-293, yololayer.cu:339-360)
int32_t yolo_create(void*, const char*, const trtx_plugin_field* f, int32_t nb, trtx_plugin_vtbl* out) {
    // the anchor-based plugin shares the registered name: told apart by its two fields "netinfo" + "kernels"
    if (nb == 2 && f && f[0].name && f[1].name && !strcmp(f[0].name, "netinfo") && !strcmp(f[1].name, "kernels")) return yolo5_create(f, out);
    if (nb != 1 || !f || !f[0].name || strcmp(f[0].name, "combinedInfo") != 0 || f[0].length < 10) return 1;
    const int* ci = static_cast<const int*>(f[0].data);
    auto* y = new YoloLayer();
    y->class_count = ci[0];
    y->n_kpt = ci[1];
    y->kpt_conf = (float)ci[2];
    y->net_w = ci[3];
    y->net_h = ci[4];
    y->max_out = ci[5];
    y->seg = ci[6] != 0;
    y->pose = ci[7] != 0;
    y->obb = ci[8] != 0;
    y->strides.assign(ci + 9, ci + f[0].length);
    yolo_fill(out, y);
    return 0;
}
int32_t yolo_deserialize(void*, const char*, const void* data, size_t len, trtx_plugin_vtbl* out) {
    YoloLayer* y = YoloLayer::from_blob(data, len);
    if (y) {
        yolo_fill(out, y);
        return 0;
    }
    // not the YOLOv8 layout (both parsers check the exact length): the anchor-based blob
    Yolo5Layer* y5 = Yolo5Layer::from_blob(data, len);
    if (!y5) return 1;
    yolo5_fill(out, y5);
    return 0;
}
const char* yolo_creator_name(void*) { return "YoloLayer_TRT"; }
const char* yolo_creator_version(void*) { return "1"; }

// ------------------------------------------------------------------------------------------------
// "Decode_TRT"/"1" — retinaface/decode.{h,cu}.  The reference fixes INPUT_H/W at compile time and serializes
// nothing (decode.cu:19-26); here the network size is learnt in configurePlugin from the stride-8 input
// (32, H/8, W/8) and stored in the blob as two ints so 1280x1280 engines deserialize without recompiling
// (SURVEY.md Appendix A.9).  An empty blob means the reference's 480x640.
struct RetinaDecode {
    int net_h = 480, net_w = 640;
};
void rdec_fill(trtx_plugin_vtbl* v, RetinaDecode* d);
int32_t rdec_nb_outputs(void*) { return 1; }
int32_t rdec_output_dims(void*, int32_t, const trtx_dims* in, int32_t nb, trtx_dims* out) {
    if (nb != 3 || in[0].nb != 3) return 1;
    const int64_t h = in[0].d[1] * 8, w = in[0].d[2] * 8;
    out->nb = 3;  // Dims3(totalCount, 1, 1), decode.cu:33-42
    out->d[0] = (int64_t)trtx_retina_decode_output_floats((int)h, (int)w);
    out->d[1] = 1;
    out->d[2] = 1;
    return 0;
}
int32_t rdec_configure(void* s, const trtx_dims* in, int32_t nb_in, const trtx_dims*, int32_t, int32_t) {
    auto* d = static_cast<RetinaDecode*>(s);
    if (nb_in != 3) return 1;
    d->net_h = (int)in[0].d[1] * 8;
    d->net_w = (int)in[0].d[2] * 8;
    for (int l = 0; l < 3; ++l)
        if (in[l].nb != 3 || in[l].d[0] != 32 || in[l].d[1] != d->net_h / (8 << l) || in[l].d[2] != d->net_w / (8 << l)) return 1;
    return 0;
}
int32_t rdec_initialize(void*) { return 0; }
void rdec_terminate(void*) {}
size_t rdec_workspace(void* s, int32_t max_batch) {
    auto* d = static_cast<RetinaDecode*>(s);
    return trtx_retina_decode_workspace(max_batch, d->net_h, d->net_w);
}
int32_t rdec_enqueue(void* s, int32_t batch, const void* const* inputs, void* const* outputs, void* ws, trtx_stream_t stream) {
    auto* d = static_cast<RetinaDecode*>(s);
    return trtx_retina_decode(reinterpret_cast<const float* const*>(inputs), batch, d->net_h, d->net_w,
                              static_cast<float*>(outputs[0]), ws, trtx_retina_decode_workspace(batch, d->net_h, d->net_w), stream);
}
size_t rdec_ser_size(void*) { return 2 * sizeof(int); }
void rdec_serialize(void* s, void* buf) {
    auto* d = static_cast<RetinaDecode*>(s);
    const int v[2] = {d->net_h, d->net_w};
    memcpy(buf, v, sizeof v);
}
const char* rdec_type(void*) { return "Decode_TRT"; }
const char* rdec_version(void*) { return "1"; }
int32_t rdec_clone(void* s, trtx_plugin_vtbl* out) {
    rdec_fill(out, new RetinaDecode(*static_cast<RetinaDecode*>(s)));
    return 0;
}
void rdec_destroy(void* s) { delete static_cast<RetinaDecode*>(s); }
void rdec_fill(trtx_plugin_vtbl* v, RetinaDecode* d) {
    v->self = d;
    v->get_nb_outputs = rdec_nb_outputs;
    v->get_output_dims = rdec_output_dims;
    v->configure = rdec_configure;
    v->initialize = rdec_initialize;
    v->terminate = rdec_terminate;
    v->workspace_size = rdec_workspace;
    v->enqueue = rdec_enqueue;
    v->serialization_size = rdec_ser_size;
    v->serialize = rdec_serialize;
    v->plugin_type = rdec_type;
    v->plugin_version = rdec_version;
    v->clone = rdec_clone;
    v->destroy = rdec_destroy;
}
int32_t rdec_create(void*, const char*, const trtx_plugin_field*, int32_t, trtx_plugin_vtbl* out) {  // empty field collection, retina_r50.cpp:204-206
    rdec_fill(out, new RetinaDecode());
    return 0;
}
int32_t rdec_deserialize(void*, const char*, const void* data, size_t len, trtx_plugin_vtbl* out) {
    auto* d = new RetinaDecode();
    if (len == 2 * sizeof(int)) {
        int v[2];
        memcpy(v, data, sizeof v);
        d->net_h = v[0];
        d->net_w = v[1];
    } else if (len != 0) {
        delete d;
        return 1;
    }
    rdec_fill(out, d);
    return 0;
}
const char* rdec_creator_name(void*) { return "Decode_TRT"; }
const char* rdec_creator_version(void*) { return "1"; }

// ------------------------------------------------------------------------------------------------------------------
// "Mish_TRT"/"1" - yolov4/mish.{h,cu} (the same plugin ships with yolov5-v1..v3 era samples and scaled-yolov4): one input of any CHW
// shape, output of the same shape, out = x * tanh(softplus(x)) with the reference's threshold-20 softplus (mish.cu:113-135).
// Blob = int input_size_ (elements per sample, mish.cu:18-32), learnt in getOutputDimensions (mish.cu:40-47).
// Inside an engine the lowering pass never calls enqueue: Conv -> Scale -> Mish_TRT becomes the convolution's epilogue.
struct Mish {
    int input_size = 0;
};
void mish_fill(trtx_plugin_vtbl* v, Mish* m);
int32_t mish_nb_outputs(void*) { return 1; }
int32_t mish_output_dims(void* s, int32_t idx, const trtx_dims* in, int32_t nb, trtx_dims* out) {
    if (nb != 1 || idx != 0 || in[0].nb < 1) return 1;
    int64_t n = 1;
    for (int k = 0; k < in[0].nb; ++k) n *= in[0].d[k];
    static_cast<Mish*>(s)->input_size = (int)n;
    *out = in[0];
    return 0;
}
int32_t mish_configure(void* s, const trtx_dims* in, int32_t nb_in, const trtx_dims*, int32_t, int32_t) {
    if (nb_in != 1) return 1;
    int64_t n = 1;
    for (int k = 0; k < in[0].nb; ++k) n *= in[0].d[k];
    static_cast<Mish*>(s)->input_size = (int)n;
    return 0;
}
int32_t mish_initialize(void*) { return 0; }
void mish_terminate(void*) {}
size_t mish_workspace(void*, int32_t) { return 0; }
int32_t mish_enqueue(void* s, int32_t batch, const void* const* inputs, void* const* outputs, void*, trtx_stream_t stream) {
    return trtx_mish(static_cast<const float*>(inputs[0]), static_cast<float*>(outputs[0]), (size_t)static_cast<Mish*>(s)->input_size * (size_t)batch,
                     stream);
}
size_t mish_ser_size(void*) { return sizeof(int); }
void mish_serialize(void* s, void* buf) { memcpy(buf, &static_cast<Mish*>(s)->input_size, sizeof(int)); }
const char* mish_type(void*) { return "Mish_TRT"; }
const char* mish_version(void*) { return "1"; }
int32_t mish_clone(void* s, trtx_plugin_vtbl* out) {
    mish_fill(out, new Mish(*static_cast<Mish*>(s)));
    return 0;
}
void mish_destroy(void* s) { delete static_cast<Mish*>(s); }
void mish_fill(trtx_plugin_vtbl* v, Mish* m) {
    v->self = m;
    v->get_nb_outputs = mish_nb_outputs;
    v->get_output_dims = mish_output_dims;
    v->configure = mish_configure;
    v->initialize = mish_initialize;
    v->terminate = mish_terminate;
    v->workspace_size = mish_workspace;
    v->enqueue = mish_enqueue;
    v->serialization_size = mish_ser_size;
    v->serialize = mish_serialize;
    v->plugin_type = mish_type;
    v->plugin_version = mish_version;
    v->clone = mish_clone;
    v->destroy = mish_destroy;
}
int32_t mish_create(void*, const char*, const trtx_plugin_field*, int32_t, trtx_plugin_vtbl* out) {  // empty field collection, mish.cu:166-178
    mish_fill(out, new Mish());
    return 0;
}
int32_t mish_deserialize(void*, const char*, const void* data, size_t len, trtx_plugin_vtbl* out) {  // mish.cu:18-22
    if (len != sizeof(int)) return 1;
    auto* m = new Mish();
    memcpy(&m->input_size, data, sizeof(int));
    mish_fill(out, m);
    return 0;
}
const char* mish_creator_name(void*) { return "Mish_TRT"; }
const char* mish_creator_version(void*) { return "1"; }

}  // namespace

bool builtin_is_mish(const trtx_plugin_vtbl& v) { return v.enqueue == mish_enqueue && v.self; }

bool builtin_yolo_params(const trtx_plugin_vtbl& v, YoloLayerParams* out) {
    if (v.enqueue != yolo_enqueue || !v.self) return false;
    const auto* y = static_cast<const YoloLayer*>(v.self);
    out->classes = y->class_count;
    out->net_w = y->net_w;
    out->net_h = y->net_h;
    out->max_out = y->max_out;
    out->strides = y->strides;
    out->det_only = !(y->seg || y->pose || y->obb);
    return true;
}

// built-in plugins only enqueue kernels and stream-ordered memsets on the caller's stream: safe inside a stream capture
bool builtin_plugin_capturable(const trtx_plugin_vtbl& v) {
    return v.enqueue == yolo_enqueue || v.enqueue == yolo5_enqueue || v.enqueue == rdec_enqueue || v.enqueue == mish_enqueue;
}

void register_builtin_plugins(PluginRegistry& r) {
    trtx_creator_vtbl c{};
    c.plugin_name = yolo_creator_name;
    c.plugin_version = yolo_creator_version;
    c.create = yolo_create;
    c.deserialize = yolo_deserialize;
    r.add(c);
    trtx_creator_vtbl d{};
    d.plugin_name = rdec_creator_name;
    d.plugin_version = rdec_creator_version;
    d.create = rdec_create;
    d.deserialize = rdec_deserialize;
    r.add(d);
    trtx_creator_vtbl m{};
    m.plugin_name = mish_creator_name;
    m.plugin_version = mish_creator_version;
    m.create = mish_create;
    m.deserialize = mish_deserialize;
    r.add(m);
}

}  // namespace trtx
