#include "graph.h"

#include <math.h>
#include <string.h>

#include <array>
#include <sstream>

#include "plugin.h"

namespace trtx {

int Network::add_tensor(const Dims& d, int dtype, int producer, int slot) {
    TensorDef t;
    t.id = (int)tensors.size();
    t.dims = d;
    t.dtype = dtype;
    t.producer = producer;
    t.producer_slot = slot;
    t.name = "(Unnamed Tensor " + std::to_string(t.id) + ")";
    tensors.push_back(t);
    return t.id;
}

int Network::add_input(const char* name, int dtype, const Dims& d) {
    const int id = add_tensor(d, dtype, -1, 0);
    tensors[id].name = name ? name : "";
    tensors[id].is_input = true;
    return id;
}

int Network::add_layer(LayerDef&& l) {
    for (int in : l.inputs)
        if (in < 0 || in >= (int)tensors.size()) {
            error = "layer input tensor id out of range";
            return -1;
        }
    const int idx = (int)layers.size();
    int n_out = 1;
    if (l.kind == L_PLUGIN) {
        if (!l.plugin) {
            error = "plugin layer without plugin";
            return -1;
        }
        n_out = l.plugin->nb_outputs();
    }
    if (l.name.empty()) l.name = "(Unnamed Layer* " + std::to_string(idx) + ")";
    layers.push_back(std::move(l));
    for (int s = 0; s < n_out; ++s) layers[idx].outputs.push_back(add_tensor(Dims{}, TRTX_DTYPE_FLOAT, idx, s));
    // TensorRT validates at build time, not at add time: parameters such as stride/padding arrive through
    // setters after the layer exists (e.g. SPPF's 5x5 pool on a 4x4 map is only valid once padding is set).
    // A failed inference leaves the outputs shapeless and is re-checked by validate().
    if (!infer(idx))
        for (int t : layers[idx].outputs) tensors[t].dims = Dims{};
    return idx;
}

bool Network::validate() {
    error.clear();
    for (size_t i = 0; i < layers.size(); ++i)
        if (!infer((int)i)) return false;
    return true;
}

bool Network::mark_output(int tensor) {
    if (tensor < 0 || tensor >= (int)tensors.size()) return false;
    tensors[tensor].is_output = true;
    return true;
}

int Network::find_tensor(const std::string& name) const {
    for (const auto& t : tensors)
        if (t.name == name) return t.id;
    return -1;
}

std::vector<int> Network::input_ids() const {
    std::vector<int> v;
    for (const auto& t : tensors)
        if (t.is_input) v.push_back(t.id);
    return v;
}

std::vector<int> Network::output_ids() const {
    std::vector<int> v;
    for (const auto& t : tensors)
        if (t.is_output) v.push_back(t.id);
    return v;
}

static bool fail(Network* n, const std::string& m) {
    n->error = m;
    return false;
}

// index of the channel dimension of an "image" tensor: CHW (implicit) or NCHW (explicit)
static int chan_axis(const Network& n, const Dims& d) {
    return (n.explicit_batch && d.nb >= 4) ? 1 : (d.nb >= 3 ? d.nb - 3 : 0);
}

bool Network::infer(int li) {
    LayerDef& l = layers[li];
    auto in = [&](int i) -> const Dims& { return tensors[l.inputs[i]].dims; };
    auto set_out = [&](int slot, const Dims& d) { tensors[l.outputs[slot]].dims = d; };
    switch (l.kind) {
        case L_CONV:
        case L_DECONV:
        case L_POOLING: {
            const Dims& x = in(0);
            if (x.nb < 3) return fail(this, l.name + ": needs a CHW tensor");
            Dims o = x;
            const int c = x.nb - 3, h = x.nb - 2;
            for (int a = 0; a < 2; ++a)
                if (l.stride[a] < 1 || l.kernel[a] < 1 || l.dilation[a] < 1 || l.padding[a] < 0) return fail(this, l.name + ": stride, kernel and dilation must be >= 1, padding >= 0");
            if ((l.kind == L_CONV || l.kind == L_DECONV) && l.groups < 1) return fail(this, l.name + ": groups must be >= 1");
            for (int a = 0; a < 2; ++a) {
                const int64_t sz = x.d[h + a];
                int64_t r;
                if (l.kind == L_DECONV)
                    r = (sz - 1) * l.stride[a] - 2 * l.padding[a] + (int64_t)l.dilation[a] * (l.kernel[a] - 1) + 1;
                else if (l.kind == L_CONV)
                    r = (sz + 2 * l.padding[a] - (int64_t)l.dilation[a] * (l.kernel[a] - 1) - 1) / l.stride[a] + 1;
                else
                    r = (sz + 2 * l.padding[a] - l.kernel[a]) / l.stride[a] + 1;
                if (r < 1) return fail(this, l.name + ": empty output");
                o.d[h + a] = r;
            }
            if (l.kind != L_POOLING) {
                if (l.groups < 1 || x.d[c] % l.groups || l.nb_out % l.groups)
                    return fail(this, l.name + ": channels not divisible by groups");
                const int64_t per = (l.kind == L_CONV) ? (x.d[c] / l.groups) * l.nb_out : (l.nb_out / l.groups) * x.d[c];
                if ((int64_t)l.w0.size() != per * l.kernel[0] * l.kernel[1])
                    return fail(this, l.name + ": kernel weight count mismatch (" + std::to_string(l.w0.size()) + ")");
                if (!l.w1.empty() && (int64_t)l.w1.size() != l.nb_out) return fail(this, l.name + ": bias count");
                o.d[c] = l.nb_out;
            }
            set_out(0, o);
            return true;
        }
        case L_FULLY_CONNECTED: {
            const Dims& x = in(0);
            if (x.nb < 3) return fail(this, l.name + ": FC needs CHW input");
            const int64_t k = x.d[x.nb - 3] * x.d[x.nb - 2] * x.d[x.nb - 1];
            if ((int64_t)l.w0.size() != k * l.nb_out) return fail(this, l.name + ": FC weight count mismatch");
            Dims o = x;
            o.d[x.nb - 3] = l.nb_out;
            o.d[x.nb - 2] = 1;
            o.d[x.nb - 1] = 1;
            set_out(0, o);
            return true;
        }
        case L_ACTIVATION:
        case L_IDENTITY:
        case L_SOFTMAX:
            set_out(0, in(0));
            return true;
        case L_SCALE: {
            const Dims& x = in(0);
            const int64_t c = l.op == TRTX_SCALE_CHANNEL ? x.d[chan_axis(*this, x)] : (l.op == TRTX_SCALE_UNIFORM ? 1 : x.volume());
            for (const auto* w : {&l.w0, &l.w1, &l.w2})
                if (!w->empty() && (int64_t)w->size() != c) return fail(this, l.name + ": scale weight count");
            set_out(0, x);
            return true;
        }
        case L_ELEMENTWISE: {
            const Dims &a = in(0), &b = in(1);
            if (a.nb != b.nb) return fail(this, l.name + ": elementwise rank mismatch");
            Dims o = a;
            for (int i = 0; i < a.nb; ++i) {
                if (a.d[i] != b.d[i] && a.d[i] != 1 && b.d[i] != 1) return fail(this, l.name + ": elementwise dims");
                o.d[i] = a.d[i] > b.d[i] ? a.d[i] : b.d[i];
            }
            set_out(0, o);
            return true;
        }
        case L_CONCAT: {
            const Dims& a = in(0);
            int ax = l.axis;
            if (ax < 0) ax = (explicit_batch && a.nb >= 4) ? 1 : (a.nb >= 3 ? a.nb - 3 : 0);
            Dims o = a;
            o.d[ax] = 0;
            for (size_t i = 0; i < l.inputs.size(); ++i) {
                const Dims& x = in((int)i);
                if (x.nb != a.nb) return fail(this, l.name + ": concat rank mismatch");
                for (int k = 0; k < a.nb; ++k)
                    if (k != ax && x.d[k] != a.d[k]) return fail(this, l.name + ": concat dims mismatch");
                o.d[ax] += x.d[ax];
            }
            l.axis = ax;
            set_out(0, o);
            return true;
        }
        case L_SLICE: {
            const Dims& x = in(0);
            if (l.start.nb != x.nb || l.size.nb != x.nb || l.step.nb != x.nb) return fail(this, l.name + ": slice rank");
            for (int i = 0; i < x.nb; ++i)
                if (l.start.d[i] < 0 || l.size.d[i] < 1 || l.start.d[i] + (l.size.d[i] - 1) * l.step.d[i] >= x.d[i])
                    return fail(this, l.name + ": slice out of range");
            set_out(0, l.size);
            return true;
        }
        case L_SHUFFLE: {
            const Dims& x = in(0);
            Dims t = x;
            for (int i = 0; i < x.nb; ++i) {
                if (l.perm1[i] < 0 || l.perm1[i] >= x.nb) return fail(this, l.name + ": bad first transpose");
                t.d[i] = x.d[l.perm1[i]];
            }
            Dims r = t;
            if (l.reshape.nb > 0) {
                r = l.reshape;
                int64_t known = 1;
                int infer_at = -1;
                for (int i = 0; i < r.nb; ++i) {
                    if (r.d[i] == 0) r.d[i] = i < t.nb ? t.d[i] : 1;
                    if (r.d[i] == -1)
                        infer_at = i;
                    else
                        known *= r.d[i];
                }
                if (infer_at >= 0) {
                    if (known == 0 || t.volume() % known) return fail(this, l.name + ": cannot infer reshape dim");
                    r.d[infer_at] = t.volume() / known;
                }
                if (r.volume() != t.volume()) return fail(this, l.name + ": reshape volume mismatch");
            }
            Dims o = r;
            for (int i = 0; i < r.nb; ++i) {
                if (l.perm2[i] < 0 || l.perm2[i] >= r.nb) return fail(this, l.name + ": bad second transpose");
                o.d[i] = r.d[l.perm2[i]];
            }
            set_out(0, o);
            return true;
        }
        case L_RESIZE: {
            const Dims& x = in(0);
            Dims o = x;
            if (l.out_dims.nb > 0) {
                if (l.out_dims.nb != x.nb) return fail(this, l.name + ": resize output rank");
                o = l.out_dims;
            } else if (l.nb_scales > 0) {
                if (l.nb_scales != x.nb) return fail(this, l.name + ": resize scales rank");
                for (int i = 0; i < x.nb; ++i) o.d[i] = (int64_t)floor((double)x.d[i] * (double)l.scales[i]);
            }
            set_out(0, o);
            return true;
        }
        case L_MATMUL: {
            Dims a = in(0), b = in(1);
            if (a.nb < 2 || b.nb < 2 || a.nb != b.nb) return fail(this, l.name + ": matmul needs equal ranks >= 2");
            const int n = a.nb;
            int64_t am = a.d[n - 2], ak = a.d[n - 1], bk = b.d[n - 2], bn = b.d[n - 1];
            if (l.mm_op[0] == TRTX_MATMUL_TRANSPOSE) std::swap(am, ak);
            if (l.mm_op[1] == TRTX_MATMUL_TRANSPOSE) std::swap(bk, bn);
            if (ak != bk) return fail(this, l.name + ": matmul inner dims mismatch");
            Dims o = a;
            for (int i = 0; i < n - 2; ++i) {
                if (a.d[i] != b.d[i] && a.d[i] != 1 && b.d[i] != 1) return fail(this, l.name + ": matmul batch dims");
                o.d[i] = a.d[i] > b.d[i] ? a.d[i] : b.d[i];
            }
            o.d[n - 2] = am;
            o.d[n - 1] = bn;
            set_out(0, o);
            return true;
        }
        case L_CONSTANT: {
            if ((int64_t)l.w0.size() != l.out_dims.volume()) return fail(this, l.name + ": constant count mismatch");
            set_out(0, l.out_dims);
            return true;
        }
        case L_REDUCE: {
            const Dims& x = in(0);
            Dims o;
            for (int i = 0; i < x.nb; ++i) {
                const bool red = (l.axis >> i) & 1;
                if (red) {
                    if (l.keep_dims) o.d[o.nb++] = 1;
                } else {
                    o.d[o.nb++] = x.d[i];
                }
            }
            set_out(0, o);
            return true;
        }
        case L_PLUGIN: {
            std::vector<Dims> ins;
            for (size_t i = 0; i < l.inputs.size(); ++i) ins.push_back(in((int)i));
            for (size_t s = 0; s < l.outputs.size(); ++s) {
                Dims o;
                if (!l.plugin->output_dims((int)s, ins, &o)) return fail(this, l.name + ": plugin getOutputDimensions failed");
                set_out((int)s, o);
            }
            return true;
        }
        default:
            return fail(this, "unknown layer kind");
    }
}

// ---------------------------------------------------------------------------------------------------------
// binary (de)serialisation
namespace {
struct Writer {
    std::vector<uint8_t>& b;
    void raw(const void* p, size_t n) {
        const uint8_t* c = static_cast<const uint8_t*>(p);
        b.insert(b.end(), c, c + n);
    }
    template <typename T>
    void pod(const T& v) {
        raw(&v, sizeof(T));
    }
    void str(const std::string& s) {
        pod<uint32_t>((uint32_t)s.size());
        raw(s.data(), s.size());
    }
    void dims(const Dims& d) {
        pod<int32_t>(d.nb);
        for (int i = 0; i < 8; ++i) pod<int64_t>(d.d[i]);
    }
    void ivec(const std::vector<int>& v) {
        pod<uint32_t>((uint32_t)v.size());
        for (int x : v) pod<int32_t>(x);
    }
    // float blobs are 16-byte aligned inside the plan so they can be read in place
    size_t fvec(const std::vector<float>& v) {
        pod<uint64_t>((uint64_t)v.size());
        while (b.size() % 16) b.push_back(0);
        const size_t off = b.size();
        raw(v.data(), v.size() * sizeof(float));
        return off;
    }
};

struct Reader {
    const uint8_t* p;
    size_t n, pos = 0;
    bool ok = true;
    bool raw(void* o, size_t k) {
        if (!ok || pos + k > n) return ok = false;
        memcpy(o, p + pos, k);
        pos += k;
        return true;
    }
    template <typename T>
    T pod() {
        T v{};
        raw(&v, sizeof(T));
        return v;
    }
    std::string str() {
        const uint32_t k = pod<uint32_t>();
        if (!ok || pos + k > n) {
            ok = false;
            return {};
        }
        std::string s(reinterpret_cast<const char*>(p + pos), k);
        pos += k;
        return s;
    }
    Dims dims() {
        Dims d;
        d.nb = pod<int32_t>();
        for (int i = 0; i < 8; ++i) d.d[i] = pod<int64_t>();
        if (d.nb < 0 || d.nb > 8) ok = false;
        return d;
    }
    std::vector<int> ivec() {
        const uint32_t k = pod<uint32_t>();
        std::vector<int> v;
        for (uint32_t i = 0; ok && i < k; ++i) v.push_back(pod<int32_t>());
        return v;
    }
    std::vector<float> fvec(size_t* off_out = nullptr) {
        const uint64_t k = pod<uint64_t>();
        while (pos % 16) ++pos;
        if (off_out) *off_out = pos;
        if (!ok || pos > n || k > (n - pos) / 4) {  // k * 4 must not wrap
            ok = false;
            return {};
        }
        std::vector<float> v(k);
        memcpy(v.data(), p + pos, k * 4);
        pos += k * 4;
        return v;
    }
};

const char kMagic[8] = {'T', 'R', 'T', 'X', 'P', 'L', 'N', '1'};
}  // namespace

void Network::serialize(std::vector<uint8_t>& out, std::vector<std::array<size_t, 3>>* w_offsets) const {
    Writer w{out};
    if (w_offsets) w_offsets->assign(layers.size(), {0, 0, 0});
    w.raw(kMagic, 8);
    w.pod<uint32_t>(4);  // format version (2: + int8 flag and per-tensor calibration scales; 3: + max_aux_streams; 4: + kernel tactics)
    w.pod<uint8_t>(explicit_batch);
    w.pod<uint8_t>(fp16);
    w.pod<uint8_t>(int8);
    w.pod<uint32_t>((uint32_t)tensor_scale.size());
    for (float sc : tensor_scale) w.pod<float>(sc);
    w.pod<int32_t>(max_aux_streams);
    w.pod<uint8_t>(tactics_timed);
    w.pod<uint32_t>((uint32_t)tactics.size());
    for (const TacticEntry& t : tactics) w.raw(&t, sizeof t);
    w.pod<int32_t>(max_batch);
    w.pod<uint32_t>((uint32_t)tensors.size());
    for (const auto& t : tensors) {
        w.str(t.name);
        w.dims(t.dims);
        w.pod<int32_t>(t.dtype);
        w.pod<int32_t>(t.producer);
        w.pod<int32_t>(t.producer_slot);
        w.pod<uint8_t>(t.is_input);
        w.pod<uint8_t>(t.is_output);
    }
    w.pod<uint32_t>((uint32_t)layers.size());
    for (size_t li = 0; li < layers.size(); ++li) {
        const auto& l = layers[li];
        w.pod<int32_t>(l.kind);
        w.str(l.name);
        w.ivec(l.inputs);
        w.ivec(l.outputs);
        w.pod(l.nb_out);
        w.pod(l.kernel);
        w.pod(l.stride);
        w.pod(l.padding);
        w.pod(l.dilation);
        w.pod(l.groups);
        w.pod(l.op);
        w.pod(l.alpha);
        w.pod(l.beta);
        w.pod(l.axis);
        w.pod(l.keep_dims);
        w.pod(l.avg_exclusive);
        w.pod(l.mm_op);
        w.dims(l.reshape);
        w.pod(l.perm1);
        w.pod(l.perm2);
        w.dims(l.start);
        w.dims(l.size);
        w.dims(l.step);
        w.pod(l.scales);
        w.pod(l.nb_scales);
        w.dims(l.out_dims);
        const size_t o0 = w.fvec(l.w0), o1 = w.fvec(l.w1), o2 = w.fvec(l.w2);
        if (w_offsets) (*w_offsets)[li] = {o0, o1, o2};
        if (l.kind == L_PLUGIN) {
            w.str(l.plugin->type());
            w.str(l.plugin->version());
            std::vector<uint8_t> blob = l.plugin->serialize();
            w.pod<uint64_t>(blob.size());
            w.raw(blob.data(), blob.size());
        }
    }
}

std::unique_ptr<Network> Network::deserialize(const uint8_t* data, size_t size, std::string* err) {
    Reader r{data, size};
    char magic[8];
    r.raw(magic, 8);
    if (!r.ok || memcmp(magic, kMagic, 8) != 0) {
        if (err) *err = "not a trtx plan (bad magic)";
        return nullptr;
    }
    const uint32_t version = r.pod<uint32_t>();
    if (version < 1 || version > 4) {
        if (err) *err = "unsupported plan version";
        return nullptr;
    }
    const bool eb = r.pod<uint8_t>();
    std::unique_ptr<Network> n(new Network(eb ? 1u : 0u));
    n->fp16 = r.pod<uint8_t>();
    if (version >= 2) {
        n->int8 = r.pod<uint8_t>();
        const uint32_t ns = r.pod<uint32_t>();
        if (!r.ok || ns > (r.n - r.pos) / 4) {
            if (err) *err = "truncated or corrupt plan";
            return nullptr;
        }
        for (uint32_t i = 0; i < ns; ++i) n->tensor_scale.push_back(r.pod<float>());
    }
    if (version >= 3) {
        n->max_aux_streams = r.pod<int32_t>();
        if (n->max_aux_streams < -1 || n->max_aux_streams > 15) {
            if (err) *err = "corrupt plan (max_aux_streams)";
            return nullptr;
        }
    }
    if (version >= 4) {
        n->tactics_timed = r.pod<uint8_t>() != 0;
        const uint32_t ntac = r.pod<uint32_t>();
        if (!r.ok || ntac > (r.n - r.pos) / sizeof(TacticEntry)) {
            if (err) *err = "truncated or corrupt plan";
            return nullptr;
        }
        n->tactics.resize(ntac);
        for (uint32_t i = 0; i < ntac; ++i) r.raw(&n->tactics[i], sizeof(TacticEntry));
    }
    n->max_batch = r.pod<int32_t>();
    const uint32_t nt = r.pod<uint32_t>();
    for (uint32_t i = 0; r.ok && i < nt; ++i) {
        TensorDef t;
        t.id = (int)i;
        t.name = r.str();
        t.dims = r.dims();
        t.dtype = r.pod<int32_t>();
        t.producer = r.pod<int32_t>();
        t.producer_slot = r.pod<int32_t>();
        t.is_input = r.pod<uint8_t>();
        t.is_output = r.pod<uint8_t>();
        n->tensors.push_back(t);
    }
    const uint32_t nl = r.pod<uint32_t>();
    for (uint32_t i = 0; r.ok && i < nl; ++i) {
        LayerDef l;
        l.kind = r.pod<int32_t>();
        l.name = r.str();
        l.inputs = r.ivec();
        l.outputs = r.ivec();
        r.raw(&l.nb_out, sizeof l.nb_out);
        r.raw(l.kernel, sizeof l.kernel);
        r.raw(l.stride, sizeof l.stride);
        r.raw(l.padding, sizeof l.padding);
        r.raw(l.dilation, sizeof l.dilation);
        r.raw(&l.groups, sizeof l.groups);
        r.raw(&l.op, sizeof l.op);
        r.raw(&l.alpha, sizeof l.alpha);
        r.raw(&l.beta, sizeof l.beta);
        r.raw(&l.axis, sizeof l.axis);
        r.raw(&l.keep_dims, sizeof l.keep_dims);
        r.raw(&l.avg_exclusive, sizeof l.avg_exclusive);
        r.raw(l.mm_op, sizeof l.mm_op);
        l.reshape = r.dims();
        r.raw(l.perm1, sizeof l.perm1);
        r.raw(l.perm2, sizeof l.perm2);
        l.start = r.dims();
        l.size = r.dims();
        l.step = r.dims();
        r.raw(l.scales, sizeof l.scales);
        r.raw(&l.nb_scales, sizeof l.nb_scales);
        l.out_dims = r.dims();
        l.w0 = r.fvec();
        l.w1 = r.fvec();
        l.w2 = r.fvec();
        if (r.ok && l.kind == L_PLUGIN) {
            const std::string type = r.str(), ver = r.str();
            const uint64_t len = r.pod<uint64_t>();
            if (!r.ok || r.pos > r.n || len > r.n - r.pos) {
                r.ok = false;
                break;
            }
            l.plugin = PluginRegistry::instance().deserialize(type, ver, r.p + r.pos, len);
            r.pos += len;
            if (!l.plugin) {
                if (err) *err = "no plugin creator registered for " + type + "/" + ver;
                return nullptr;
            }
        }
        for (int t : l.inputs)
            if (t < 0 || t >= (int)n->tensors.size()) r.ok = false;
        for (int t : l.outputs)
            if (t < 0 || t >= (int)n->tensors.size()) r.ok = false;
        n->layers.push_back(std::move(l));
    }
    if (!r.ok) {
        if (err) *err = "truncated or corrupt plan";
        return nullptr;
    }
    // a plan is data from outside: re-run shape inference so that weight counts, strides and dims are consistent before anything
    // sizes a buffer from them
    if (!n->validate()) {
        if (err) *err = "plan fails validation: " + n->error;
        return nullptr;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------------
static void json_str(std::ostringstream& o, const std::string& s) {
    o << '"';
    for (char c : s) {
        if (c == '"' || c == '\\')
            o << '\\' << c;
        else if ((unsigned char)c < 0x20)
            o << ' ';
        else
            o << c;
    }
    o << '"';
}
static void json_dims(std::ostringstream& o, const Dims& d) {
    o << '[';
    for (int i = 0; i < d.nb; ++i) o << (i ? "," : "") << d.d[i];
    o << ']';
}
template <typename T>
static void json_arr(std::ostringstream& o, const T* v, int n) {
    o << '[';
    for (int i = 0; i < n; ++i) o << (i ? "," : "") << v[i];
    o << ']';
}

std::string Network::describe_json() const {
    // serialisation is deterministic: re-serialise to learn where each weight vector lives in the plan
    std::vector<std::array<size_t, 3>> offs;
    {
        std::vector<uint8_t> tmp;
        serialize(tmp, &offs);
    }
    std::ostringstream o;
    o << "{\"explicit_batch\":" << (explicit_batch ? "true" : "false") << ",\"fp16\":" << (fp16 ? "true" : "false") << ",\"int8\":" << (int8 ? "true" : "false") << ",\"max_aux_streams\":" << max_aux_streams
      << ",\"max_batch\":" << max_batch << ",\"tactics_timed\":" << (tactics_timed ? "true" : "false") << ",\"tactics\":" << tactics.size() << ",\"tensors\":[";
    for (size_t i = 0; i < tensors.size(); ++i) {
        const auto& t = tensors[i];
        o << (i ? "," : "") << "{\"id\":" << t.id << ",\"name\":";
        json_str(o, t.name);
        o << ",\"dims\":";
        json_dims(o, t.dims);
        o << ",\"is_input\":" << (t.is_input ? "true" : "false") << ",\"is_output\":" << (t.is_output ? "true" : "false")
          << "}";
    }
    o << "],\"layers\":[";
    for (size_t i = 0; i < layers.size(); ++i) {
        const auto& l = layers[i];
        o << (i ? "," : "") << "{\"kind\":" << l.kind << ",\"name\":";
        json_str(o, l.name);
        o << ",\"inputs\":";
        json_arr(o, l.inputs.data(), (int)l.inputs.size());
        o << ",\"outputs\":";
        json_arr(o, l.outputs.data(), (int)l.outputs.size());
        o << ",\"nb_out\":" << l.nb_out << ",\"kernel\":";
        json_arr(o, l.kernel, 2);
        o << ",\"stride\":";
        json_arr(o, l.stride, 2);
        o << ",\"padding\":";
        json_arr(o, l.padding, 2);
        o << ",\"dilation\":";
        json_arr(o, l.dilation, 2);
        o << ",\"groups\":" << l.groups << ",\"op\":" << l.op << ",\"alpha\":" << l.alpha << ",\"beta\":" << l.beta
          << ",\"axis\":" << l.axis << ",\"keep_dims\":" << l.keep_dims << ",\"avg_exclusive\":" << l.avg_exclusive
          << ",\"mm_op\":";
        json_arr(o, l.mm_op, 2);
        o << ",\"reshape\":";
        json_dims(o, l.reshape);
        o << ",\"has_reshape\":" << (l.reshape.nb > 0 ? "true" : "false") << ",\"perm1\":";
        json_arr(o, l.perm1, 8);
        o << ",\"perm2\":";
        json_arr(o, l.perm2, 8);
        o << ",\"start\":";
        json_dims(o, l.start);
        o << ",\"size\":";
        json_dims(o, l.size);
        o << ",\"step\":";
        json_dims(o, l.step);
        o << ",\"scales\":";
        json_arr(o, l.scales, l.nb_scales);
        o << ",\"out_dims\":";
        json_dims(o, l.out_dims);
        o << ",\"w\":[";
        const std::vector<float>* ws[3] = {&l.w0, &l.w1, &l.w2};
        for (int k = 0; k < 3; ++k) o << (k ? "," : "") << "[" << offs[i][k] << "," << ws[k]->size() << "]";
        o << "]";
        if (l.kind == L_PLUGIN) {
            o << ",\"plugin_type\":";
            json_str(o, l.plugin->type());
            const std::vector<uint8_t> pb = l.plugin->serialize();
            o << ",\"plugin_blob\":\"";
            static const char* hex = "0123456789abcdef";
            for (uint8_t c : pb) o << hex[c >> 4] << hex[c & 15];
            o << "\"";
        }
        o << "}";
    }
    o << "]}";
    return o.str();
}

}  // namespace trtx
