// Network -> Plan lowering.  See plan.h for what this stands in for.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <sstream>

#include "../options.h"
#include "pack.h"
#include "plan.h"

namespace trtx {

const char* op_kind_name(int k) {
    static const char* n[] = {"conv",     "deconv",    "pool",      "resize",     "ew_nhwc", "act_nhwc", "scale_nhwc",
                              "copy_nhwc", "reduce_hw", "to_nhwc",   "to_linear",  "gather",  "scatter",  "ew_lin",
                              "act_lin",  "scale_lin", "softmax",   "matmul",     "reduce_lin", "plugin", "copy_lin", "yolo_head",
                              "pool_chain", "depth_to_space", "roi_align", "conv_chain", "conv_group"};
    return (k >= 0 && k <= OP_CONV_GROUP) ? n[k] : "?";
}

namespace {

int act_code(int trt_type) {
    switch (trt_type) {
        case TRTX_ACTIVATION_RELU: return ACT_RELU;
        case TRTX_ACTIVATION_SIGMOID: return ACT_SIGMOID;
        case TRTX_ACTIVATION_TANH: return ACT_TANH;
        case TRTX_ACTIVATION_LEAKY_RELU: return ACT_LEAKY;
        default: return -1;
    }
}

int ew_code(int trt_op) {
    switch (trt_op) {
        case TRTX_ELEMENTWISE_SUM: return EW_SUM;
        case TRTX_ELEMENTWISE_PROD: return EW_PROD;
        case TRTX_ELEMENTWISE_MAX: return EW_MAX;
        case TRTX_ELEMENTWISE_MIN: return EW_MIN;
        case TRTX_ELEMENTWISE_SUB: return EW_SUB;
        case TRTX_ELEMENTWISE_DIV: return EW_DIV;
        case TRTX_ELEMENTWISE_POW: return EW_POW;
        default: return -1;
    }
}

struct FusedConv {
    int conv_layer = -1;
    int scale_layer = -1;
    int act1 = ACT_NONE;
    float alpha1 = 0.f;
    int residual = -1;  // network tensor id
    int act2 = ACT_NONE;
    float alpha2 = 0.f;
    int out_tensor = -1;  // network tensor the fused op produces
    int emit_at = -1;     // layer index at which the fused op is scheduled
};

struct YoloHeadFuse {
    int plugin_layer = -1;
    std::vector<int> head_tensor;  // network tensor per level: CHW (64 + classes, gh, gw)
    int dfl_conv_layer = -1;
    YoloLayerParams params;
};

struct Lowerer {
    const Network& net;
    Plan& plan;
    const Options opt = read_options();   // the environment's A/B switches as of THIS lowering (tests flip them inside one process)
    int dt;  // dtype of NHWC tensors
    std::vector<std::vector<int>> consumers;
    std::vector<int> pt_of, pt_lin, pt_nhwc;
    std::vector<bool> absorbed;
    std::vector<char> stride_folded;   // per layer: a 1x1 stride-s convolution whose producer emitted only the pixels it reads -> stride 1
    std::vector<int> group_at;
    std::vector<FusedConv> groups;
    std::vector<YoloHeadFuse> yolo_heads;
    std::vector<int> yolo_at;
    std::vector<std::pair<int, int>> aliases;  // (dst network tensor, src network tensor): dst is the same data as src
    std::string err;

    Lowerer(const Network& n, Plan& p) : net(n), plan(p) {
        dt = n.fp16 ? DT_F16 : DT_F32;
        consumers.resize(n.tensors.size());
        for (size_t li = 0; li < n.layers.size(); ++li)
            for (int t : n.layers[li].inputs) consumers[t].push_back((int)li);
        pt_of.assign(n.tensors.size(), -1);
        pt_lin.assign(n.tensors.size(), -1);
        pt_nhwc.assign(n.tensors.size(), -1);
        absorbed.assign(n.layers.size(), false);
        stride_folded.assign(n.layers.size(), 0);
        group_at.assign(n.layers.size(), -1);
        yolo_at.assign(n.layers.size(), -1);
    }

    bool fail(const std::string& m) {
        if (err.empty()) err = m;
        return false;
    }

    // image tensors: (C,H,W) per sample, or (P,C,H,W) per sample where the leading P folds into the image count
    // (TensorRT applies conv/pool/FC to the last three dims; rcnn.cpp:154-160 runs res5 on a (1000,C,14,14) tensor)
    bool spatial(const Dims& d) const { return net.explicit_batch ? d.nb == 4 : (d.nb == 3 || d.nb == 4); }

    // ---- tensors ---------------------------------------------------------------------------------
    int new_tensor(int net_t, const Dims& d, int layout, bool batched) {
        PTensor t;
        t.id = (int)plan.tensors.size();
        t.net_tensor = net_t;
        t.name = net_t >= 0 ? net.tensors[net_t].name : "";
        t.dims = d;
        t.layout = layout;
        t.batched = batched;
        if (layout == LAY_NHWC) {
            t.dtype = dt;
            if (net.explicit_batch) {
                t.nfix = (int)d.d[0];
                t.C = (int)d.d[1];
                t.H = (int)d.d[2];
                t.W = (int)d.d[3];
                t.batched = false;
            } else {
                t.C = (int)d.d[d.nb - 3];
                t.H = (int)d.d[d.nb - 2];
                t.W = (int)d.d[d.nb - 1];
                if (d.nb == 4) t.nmul = (int)d.d[0];
            }
            t.Calloc = dt == DT_F16 ? (t.C + 7) / 8 * 8 : (t.C + 3) / 4 * 4;   // a pixel's channels start on a 16-byte boundary
        } else {
            t.dtype = DT_F32;
            if (net.explicit_batch) t.batched = false;
        }
        plan.tensors.push_back(t);
        return t.id;
    }
    int new_view_nhwc(int parent, int coff, int C, int net_t) {
        PTensor t = plan.tensors[parent];
        t.id = (int)plan.tensors.size();
        t.net_tensor = net_t;
        t.name = net_t >= 0 ? net.tensors[net_t].name : "";
        t.parent = parent;
        t.coff = coff;
        t.C = C;
        t.Calloc = 0;
        t.pad_zeroed = false;
        t.dims.d[t.dims.nb - 3] = C;
        plan.tensors.push_back(t);
        return t.id;
    }
    int new_view_lin(int parent, const Dims& d, int net_t) {
        PTensor t = plan.tensors[parent];
        t.id = (int)plan.tensors.size();
        t.net_tensor = net_t;
        t.name = net_t >= 0 ? net.tensors[net_t].name : "";
        t.parent = parent;
        t.eoff = 0;
        t.dims = d;
        plan.tensors.push_back(t);
        return t.id;
    }

    POp& add_op(int kind, const std::string& name, std::vector<int> in, std::vector<int> out) {
        POp op;
        op.kind = kind;
        op.name = name;
        op.in = std::move(in);
        op.out = std::move(out);
        op.dtype = dt;
        plan.ops.push_back(std::move(op));
        return plan.ops.back();
    }

    int need_nhwc(int net_t) {
        const int p = pt_of[net_t];
        if (plan.tensors[p].layout == LAY_NHWC) return p;
        if (pt_nhwc[net_t] >= 0) return pt_nhwc[net_t];
        const int q = new_tensor(net_t, plan.tensors[p].dims, LAY_NHWC, plan.tensors[p].batched);
        plan.tensors[q].pad_zeroed = true;
        POp& op = add_op(OP_TO_NHWC, "to_nhwc:" + net.tensors[net_t].name, {p}, {q});
        op.bytes = (double)plan.tensors[p].dims.volume() * (4 + (dt == DT_F16 ? 2 : 4));
        return pt_nhwc[net_t] = q;
    }
    int need_lin(int net_t) {
        const int p = pt_of[net_t];
        if (plan.tensors[p].layout == LAY_LINEAR) return p;
        if (pt_lin[net_t] >= 0) return pt_lin[net_t];
        const int q = new_tensor(net_t, plan.tensors[p].dims, LAY_LINEAR, plan.tensors[p].batched || plan.tensors[p].nfix == 0);
        POp& op = add_op(OP_TO_LINEAR, "to_linear:" + net.tensors[net_t].name, {p}, {q});
        op.bytes = (double)plan.tensors[p].dims.volume() * (4 + (dt == DT_F16 ? 2 : 4));
        return pt_lin[net_t] = q;
    }

    // top owner of an NHWC tensor and accumulated channel offset
    int owner_of(int p, int* off) const {
        int o = 0;
        while (plan.tensors[p].parent >= 0) {
            o += plan.tensors[p].coff;
            p = plan.tensors[p].parent;
        }
        if (off) *off = o;
        return p;
    }
    bool is_binding_tensor(int p) const {
        for (int b : plan.binding_ptensor)
            if (b == p) return true;
        return false;
    }
    // try to make tensor `child` live inside `parent` at channel `coff`
    bool try_place(int child, int parent, int coff) {
        int off = 0;
        const int top = owner_of(child, &off);
        PTensor& t = plan.tensors[top];
        const PTensor& c = plan.tensors[child];
        if (t.layout != LAY_NHWC) return false;
        // only a whole, freely placeable owner may move.  Every producer writes exactly C channels (ragged channel
        // counts take the element-wise store paths), except the layout pass that zero-fills its padding.
        if (off != 0 || c.C != t.C || t.pad_zeroed || is_binding_tensor(top)) return false;
        if (owner_of(parent, nullptr) == top) return false;  // would create a cycle
        t.parent = parent;
        t.coff = coff;
        return true;
    }

    // ---- fusion analysis ----------------------------------------------------------------------------
    bool sole_consumer(int tensor, int* layer) const {
        if (consumers[tensor].size() != 1 || net.tensors[tensor].is_output) return false;
        *layer = consumers[tensor][0];
        return true;
    }

    bool is_builtin_mish(int li) const {
        const LayerDef& l = net.layers[li];
        return l.kind == L_PLUGIN && l.plugin && l.inputs.size() == 1 && l.outputs.size() == 1 && builtin_is_mish(l.plugin->v);
    }

    void analyse_fusion() {
        for (size_t li = 0; li < net.layers.size(); ++li) {
            const LayerDef& l = net.layers[li];
            if (l.kind != L_CONV && l.kind != L_FULLY_CONNECTED) continue;
            if (absorbed[li]) continue;  // already claimed (fused YOLO head)
            if (!spatial(net.tensors[l.inputs[0]].dims)) continue;
            FusedConv g;
            g.conv_layer = (int)li;
            int t = l.outputs[0];
            int last = (int)li;
            int nx;
            // Conv -> Scale (BatchNorm folded by the host code, block.cpp:45-77)
            if (sole_consumer(t, &nx) && !absorbed[nx] && net.layers[nx].kind == L_SCALE && net.layers[nx].op != TRTX_SCALE_ELEMENTWISE) {
                const LayerDef& s = net.layers[nx];
                bool pow1 = true;
                for (float p : s.w2) pow1 = pow1 && p == 1.0f;
                if (pow1) {
                    g.scale_layer = nx;
                    absorbed[nx] = true;
                    t = s.outputs[0];
                    last = std::max(last, nx);
                }
            }
            // SiLU spelled as Sigmoid + Prod (block.cpp:91-94), or a plain activation
            if (!net.tensors[t].is_output && consumers[t].size() == 2) {
                int a = consumers[t][0], b = consumers[t][1];
                if (net.layers[a].kind != L_ACTIVATION) std::swap(a, b);
                const LayerDef &la = net.layers[a], &lb = net.layers[b];
                if (la.kind == L_ACTIVATION && la.op == TRTX_ACTIVATION_SIGMOID && lb.kind == L_ELEMENTWISE &&
                    lb.op == TRTX_ELEMENTWISE_PROD && a != b && !absorbed[a] && !absorbed[b]) {
                    const int so = la.outputs[0];
                    const bool uses = (lb.inputs[0] == t && lb.inputs[1] == so) || (lb.inputs[1] == t && lb.inputs[0] == so);
                    int only;
                    if (uses && sole_consumer(so, &only) && only == b) {
                        g.act1 = ACT_SILU;
                        absorbed[a] = absorbed[b] = true;
                        t = lb.outputs[0];
                        last = std::max(last, std::max(a, b));
                    }
                }
            } else if (sole_consumer(t, &nx) && !absorbed[nx] && net.layers[nx].kind == L_ACTIVATION && act_code(net.layers[nx].op) >= 0) {
                g.act1 = act_code(net.layers[nx].op);
                g.alpha1 = net.layers[nx].alpha;
                absorbed[nx] = true;
                t = net.layers[nx].outputs[0];
                last = std::max(last, nx);
            } else if (sole_consumer(t, &nx) && !absorbed[nx] && is_builtin_mish(nx)) {
                // Conv -> Scale(BN) -> Mish_TRT (convBnMish, yolov4/yolov4.cpp:199-213): the plugin is a pointwise activation
                g.act1 = ACT_MISH;
                absorbed[nx] = true;
                t = net.layers[nx].outputs[0];
                last = std::max(last, nx);
            }
            // + residual (block.cpp:104-108 ; resnet50.cpp:146), then an optional trailing activation
            if (sole_consumer(t, &nx) && !absorbed[nx] && net.layers[nx].kind == L_ELEMENTWISE && net.layers[nx].op == TRTX_ELEMENTWISE_SUM) {
                const LayerDef& e = net.layers[nx];
                const int other = e.inputs[0] == t ? e.inputs[1] : e.inputs[0];
                if (other != t && net.tensors[other].dims == net.tensors[t].dims) {
                    g.residual = other;
                    absorbed[nx] = true;
                    t = e.outputs[0];
                    last = std::max(last, nx);
                    int n2;
                    if (sole_consumer(t, &n2) && !absorbed[n2] && net.layers[n2].kind == L_ACTIVATION && act_code(net.layers[n2].op) >= 0) {
                        g.act2 = act_code(net.layers[n2].op);
                        g.alpha2 = net.layers[n2].alpha;
                        absorbed[n2] = true;
                        t = net.layers[n2].outputs[0];
                        last = std::max(last, n2);
                    }
                }
            }
            g.out_tensor = t;
            g.emit_at = last;
            absorbed[li] = true;
            group_at[last] = (int)groups.size();
            groups.push_back(g);
        }
        // Activation applied to a concatenation of un-activated convolution outputs (RetinaFace SSH,
        // retina_r50.cpp:87-98): relu(cat(a, b, c)) == cat(relu a, relu b, relu c), so the activation moves into
        // the producers' epilogues and the concat output is used as is.
        for (size_t li = 0; li < net.layers.size(); ++li) {
            const LayerDef& l = net.layers[li];
            if (l.kind != L_CONCAT || absorbed[li]) continue;
            int nx;
            if (!sole_consumer(l.outputs[0], &nx) || absorbed[nx] || net.layers[nx].kind != L_ACTIVATION) continue;
            const int code = act_code(net.layers[nx].op);
            if (code < 0) continue;
            std::vector<int> gs;
            bool ok = true;
            for (int t : l.inputs) {
                int gi = -1;
                for (size_t k = 0; k < groups.size(); ++k)
                    if (groups[k].out_tensor == t) gi = (int)k;
                int only;
                ok = ok && gi >= 0 && groups[gi].act1 == ACT_NONE && groups[gi].residual < 0 && groups[gi].act2 == ACT_NONE &&
                     sole_consumer(t, &only) && only == (int)li;
                gs.push_back(gi);
            }
            if (!ok) continue;
            for (int gi : gs) {
                groups[gi].act1 = code;
                groups[gi].alpha1 = net.layers[nx].alpha;
            }
            absorbed[nx] = true;
            aliases.push_back({net.layers[nx].outputs[0], l.outputs[0]});
        }
    }

    // ---- YOLOv8 detect tail: flatten -> slice -> DFL(shuffle, softmax, 1x1 conv, shuffle) -> concat -> YoloLayer_TRT
    // (yolov8/src/model.cpp:263-303, block.cpp:239-257) collapses into one fused kernel when the plugin is the
    // built-in one and every intermediate tensor has no other use.
    int producer(int tensor) const { return net.tensors[tensor].producer; }
    bool only_used_by(int tensor, std::initializer_list<int> layers) const {
        if (net.tensors[tensor].is_output) return false;
        std::vector<int> want(layers), have(consumers[tensor]);
        std::sort(want.begin(), want.end());
        std::sort(have.begin(), have.end());
        return want == have;
    }
    static bool ident(const int32_t* p, int n) {
        for (int k = 0; k < n; ++k)
            if (p[k] != k) return false;
        return true;
    }
    bool match_yolo_level(int plugin_layer, int t_in, int classes, int* head, int* conv_layer, std::vector<int>* used) {
        const int lc = producer(t_in);
        if (lc < 0 || net.layers[lc].kind != L_CONCAT || net.layers[lc].inputs.size() != 2 || net.layers[lc].axis != 0) return false;
        if (!only_used_by(t_in, {plugin_layer})) return false;
        const int ta = net.layers[lc].inputs[0], tb = net.layers[lc].inputs[1];
        // box branch
        const int lsh2 = producer(ta);
        if (lsh2 < 0 || net.layers[lsh2].kind != L_SHUFFLE || !only_used_by(ta, {lc})) return false;
        const LayerDef& sh2 = net.layers[lsh2];
        const Dims& d2 = net.tensors[ta].dims;
        if (d2.nb != 2 || d2.d[0] != 4 || !ident(sh2.perm1, 3) || !ident(sh2.perm2, 2)) return false;
        const int64_t g = d2.d[1];
        const int tconv = sh2.inputs[0];
        const int lconv = producer(tconv);
        if (lconv < 0 || net.layers[lconv].kind != L_CONV || !only_used_by(tconv, {lsh2})) return false;
        const LayerDef& cv = net.layers[lconv];
        if (cv.nb_out != 1 || cv.kernel[0] != 1 || cv.kernel[1] != 1 || cv.groups != 1 || cv.stride[0] != 1 || cv.stride[1] != 1 ||
            cv.padding[0] != 0 || cv.padding[1] != 0 || cv.w0.size() != 16 || !cv.w1.empty())
            return false;
        const int tsm = cv.inputs[0];
        const int lsm = producer(tsm);
        if (lsm < 0 || net.layers[lsm].kind != L_SOFTMAX || !only_used_by(tsm, {lconv})) return false;
        if (!(net.layers[lsm].axis < 0 || net.layers[lsm].axis == 1)) return false;
        const int tsh1 = net.layers[lsm].inputs[0];
        const int lsh1 = producer(tsh1);
        if (lsh1 < 0 || net.layers[lsh1].kind != L_SHUFFLE || !only_used_by(tsh1, {lsm})) return false;
        const LayerDef& sh1 = net.layers[lsh1];
        const Dims& d1 = net.tensors[tsh1].dims;
        if (d1.nb != 3 || d1.d[0] != 16 || d1.d[1] != 4 || d1.d[2] != g) return false;
        if (!ident(sh1.perm1, 2) || sh1.reshape.nb != 3 || sh1.perm2[0] != 1 || sh1.perm2[1] != 0 || sh1.perm2[2] != 2) return false;
        const int tsa = sh1.inputs[0];
        const int lsa = producer(tsa);
        if (lsa < 0 || net.layers[lsa].kind != L_SLICE || !only_used_by(tsa, {lsh1})) return false;
        const LayerDef& sa = net.layers[lsa];
        if (sa.start.nb != 2 || sa.start.d[0] != 0 || sa.start.d[1] != 0 || sa.size.d[0] != 64 || sa.size.d[1] != g ||
            sa.step.d[0] != 1 || sa.step.d[1] != 1)
            return false;
        // class branch
        const int lsb = producer(tb);
        if (lsb < 0 || net.layers[lsb].kind != L_SLICE || !only_used_by(tb, {lc})) return false;
        const LayerDef& sb = net.layers[lsb];
        if (sb.start.nb != 2 || sb.start.d[0] != 64 || sb.start.d[1] != 0 || sb.size.d[0] != classes || sb.size.d[1] != g ||
            sb.step.d[0] != 1 || sb.step.d[1] != 1 || sb.inputs[0] != sa.inputs[0])
            return false;
        const int tflat = sa.inputs[0];
        const int lflat = producer(tflat);
        if (lflat < 0 || net.layers[lflat].kind != L_SHUFFLE || !only_used_by(tflat, {lsa, lsb})) return false;
        const LayerDef& fl = net.layers[lflat];
        const Dims& dx = net.tensors[fl.inputs[0]].dims;
        if (!ident(fl.perm1, 3) || !ident(fl.perm2, 2) || !spatial(dx) || dx.d[0] != 64 + classes || dx.d[1] * dx.d[2] != g) return false;
        *head = fl.inputs[0];
        *conv_layer = lconv;
        for (int l : {lc, lsh2, lconv, lsm, lsh1, lsa, lsb, lflat}) used->push_back(l);
        return true;
    }
    void analyse_yolo_head() {
        if (net.explicit_batch) return;   // (fp16 and, since round 5, fp32 engines: the kernel reads either element type)
        for (size_t li = 0; li < net.layers.size(); ++li) {
            const LayerDef& l = net.layers[li];
            if (l.kind != L_PLUGIN || l.outputs.size() != 1) continue;
            YoloHeadFuse f;
            if (!builtin_yolo_params(l.plugin->v, &f.params)) continue;
            if (!f.params.det_only || f.params.classes % 8 || f.params.strides.size() != l.inputs.size() || l.inputs.size() > 6) continue;
            std::vector<int> used;
            bool ok = true;
            for (size_t k = 0; ok && k < l.inputs.size(); ++k) {
                int head = -1, conv = -1;
                ok = match_yolo_level((int)li, l.inputs[k], f.params.classes, &head, &conv, &used);
                if (!ok) break;
                const Dims& dh = net.tensors[head].dims;
                ok = dh.d[1] == f.params.net_h / f.params.strides[k] && dh.d[2] == f.params.net_w / f.params.strides[k];
                if (f.dfl_conv_layer >= 0 && net.layers[conv].w0 != net.layers[f.dfl_conv_layer].w0) ok = false;
                f.dfl_conv_layer = conv;
                f.head_tensor.push_back(head);
            }
            for (int u : used) ok = ok && !absorbed[u];
            if (!ok) continue;
            for (int u : used) absorbed[u] = true;
            f.plugin_layer = (int)li;
            absorbed[li] = true;
            yolo_at[li] = (int)yolo_heads.size();
            yolo_heads.push_back(f);
        }
    }
    bool emit_yolo_head(const YoloHeadFuse& f) {
        const LayerDef& l = net.layers[f.plugin_layer];
        std::vector<int> ins;
        for (int t : f.head_tensor) ins.push_back(need_nhwc(t));
        const int out = new_tensor(l.outputs[0], net.tensors[l.outputs[0]].dims, LAY_LINEAR, true);
        POp& op = add_op(OP_YOLO_HEAD, l.name + " [fused DFL+decode]", ins, {out});
        op.src_layer = f.dfl_conv_layer;
        op.i[0] = f.params.classes;
        op.i[1] = f.params.net_h;
        op.i[2] = f.params.net_w;
        op.i[3] = f.params.max_out;
        op.i[4] = (int)f.params.strides.size();
        for (size_t k = 0; k < f.params.strides.size(); ++k) op.i[5 + k] = f.params.strides[k];
        op.ws_bytes = trtx_yolo_head_decode_workspace(plan.max_batch, f.params.net_h, f.params.net_w, f.params.strides.data(),
                                                      (int)f.params.strides.size());
        for (int t : ins) op.bytes += (double)dtype_size(dt) * plan.tensors[t].dims.volume();
        op.bytes += 4.0 * net.tensors[l.outputs[0]].dims.volume();
        pt_of[l.outputs[0]] = out;
        return true;
    }

    // ---- per-kind emission ----------------------------------------------------------------------------
    bool emit_conv(const FusedConv& g) {
        const LayerDef& l = net.layers[g.conv_layer];
        // stem: a few-channel fp32 LINEAR input (the image) feeds conv_stem directly, no layout pass
        bool stem = false;
        {
            const PTensor& src = plan.tensors[pt_of[l.inputs[0]]];
            const Dims& di = net.tensors[l.inputs[0]].dims;
            const int cin = (int)di.d[di.nb - 3];
            stem = dt == DT_F16 && l.kind == L_CONV && di.nb == (net.explicit_batch ? 4 : 3) && src.layout == LAY_LINEAR && pt_nhwc[l.inputs[0]] < 0 && cin <= 4 &&
                   g.residual < 0 && g.act2 == ACT_NONE && l.groups == 1 && l.dilation[0] == 1 && l.dilation[1] == 1 &&
                   (l.nb_out == 8 || l.nb_out == 16 || l.nb_out == 32 || l.nb_out == 64) &&
                   (size_t)l.kernel[0] * l.kernel[1] * cin * l.nb_out * 4 <= 48 * 1024;
            // fp32 engines (round 5): kernels/conv_stem_f32.hip, the same idea on the vector ALU - the layer is HBM-bound and 3x padding on the MFMA path
            if (dt == DT_F32 && opt.f32_mfma)
                stem = l.kind == L_CONV && di.nb == (net.explicit_batch ? 4 : 3) && src.layout == LAY_LINEAR && pt_nhwc[l.inputs[0]] < 0 && cin <= 4 && g.residual < 0 &&
                       g.act2 == ACT_NONE && l.groups == 1 && l.dilation[0] == 1 && l.dilation[1] == 1 && l.nb_out % 16 == 0 && l.nb_out <= 256;
        }
        const int in = stem ? pt_of[l.inputs[0]] : need_nhwc(l.inputs[0]);
        PTensor ti = plan.tensors[in];
        if (stem) {  // geometry of the LINEAR tensor viewed as an image
            const Dims& di = net.tensors[l.inputs[0]].dims;
            ti.C = (int)di.d[di.nb - 3];
            ti.H = (int)di.d[di.nb - 2];
            ti.W = (int)di.d[di.nb - 1];
        }
        const int out = new_tensor(g.out_tensor, net.tensors[g.out_tensor].dims, LAY_NHWC, true);
        int res = -1;
        if (g.residual >= 0) res = need_nhwc(g.residual);
        std::vector<int> ins = {in};
        if (res >= 0) ins.push_back(res);
        POp& op = add_op(OP_CONV, l.name, ins, {out});
        op.src_layer = g.conv_layer;
        op.scale_layer = g.scale_layer;
        op.stem = stem;
        ConvArgs& a = op.conv;
        const PTensor& to = plan.tensors[out];
        a.H = ti.H;
        a.W = ti.W;
        a.Cin = ti.C;
        a.Ho = to.H;
        a.Wo = to.W;
        a.Cout = to.C;
        if (l.kind == L_FULLY_CONNECTED) {
            a.kh = ti.H;
            a.kw = ti.W;
            a.stride_h = a.stride_w = 1;
            a.pad_h = a.pad_w = 0;
            a.dil_h = a.dil_w = 1;
            a.groups = 1;
        } else {
            a.kh = l.kernel[0];
            a.kw = l.kernel[1];
            a.stride_h = l.stride[0];
            a.stride_w = l.stride[1];
            if (stride_folded[g.conv_layer]) a.stride_h = a.stride_w = 1;   // its input was emitted at the positions it reads (RoIAlign, below)
            a.pad_h = l.padding[0];
            a.pad_w = l.padding[1];
            a.dil_h = l.dilation[0];
            a.dil_w = l.dilation[1];
            a.groups = l.groups;
        }
        a.act1 = g.act1;
        a.alpha1 = g.alpha1;
        a.act2 = g.act2;
        a.alpha2 = g.alpha2;
        op.flops = 2.0 * to.nmul * a.Ho * a.Wo * a.Cout * (double)a.kh * a.kw * (a.Cin / a.groups);
        pt_of[g.out_tensor] = out;
        return true;
    }

    bool emit_layer(int li) {
        const LayerDef& l = net.layers[li];
        auto out_dims = [&](int s = 0) -> const Dims& { return net.tensors[l.outputs[s]].dims; };
        switch (l.kind) {
            case L_CONV:
            case L_FULLY_CONNECTED:
                return fail(l.name + ": convolution on a non-image tensor is not supported");
            case L_DECONV: {
                const int in = need_nhwc(l.inputs[0]);
                const PTensor ti = plan.tensors[in];
                // kernel == stride, no padding, one group (Mask R-CNN's 2x2/2 ConvTranspose over 2048 channels,
                // rcnn.cpp:209-212): every output sub-position (r, q) is its own 1x1 convolution, so the layer is one MFMA
                // implicit GEMM with Cout' = kh*kw*Cout followed by a depth-to-space shuffle (41.7 ms -> ~0.15 ms there)
                if (dt == DT_F16 && l.groups == 1 && l.kernel[0] == l.stride[0] && l.kernel[1] == l.stride[1] && l.padding[0] == 0 &&
                    l.padding[1] == 0 && l.dilation[0] == 1 && l.dilation[1] == 1 && ti.C % 8 == 0 && l.nb_out % 8 == 0 &&
                    l.kernel[0] * l.kernel[1] > 1) {
                    Dims dmid = net.tensors[l.inputs[0]].dims;
                    dmid.d[dmid.nb - 3] = (int64_t)l.nb_out * l.kernel[0] * l.kernel[1];
                    const int mid = new_tensor(-1, dmid, LAY_NHWC, true);
                    const int out = new_tensor(l.outputs[0], out_dims(), LAY_NHWC, true);
                    {
                        POp& op = add_op(OP_CONV, l.name + " [as 1x1]", {in}, {mid});
                        op.src_layer = li;
                        op.from_deconv = true;
                        ConvArgs& a = op.conv;
                        const PTensor& tm = plan.tensors[mid];
                        a.H = ti.H; a.W = ti.W; a.Cin = ti.C; a.Ho = ti.H; a.Wo = ti.W; a.Cout = tm.C;
                        a.kh = a.kw = 1; a.stride_h = a.stride_w = 1; a.pad_h = a.pad_w = 0; a.dil_h = a.dil_w = 1; a.groups = 1;
                        op.flops = 2.0 * tm.nmul * a.Ho * a.Wo * a.Cout * (double)a.Cin;
                    }
                    POp& d2s = add_op(OP_D2S, l.name + " [depth to space]", {mid}, {out});
                    d2s.i[0] = l.kernel[0];
                    d2s.i[1] = l.kernel[1];
                    pt_of[l.outputs[0]] = out;
                    return true;
                }
                // A depthwise transposed convolution with kernel == stride, every weight 1 and no bias copies each input pixel into its
                // k x k block: a nearest-neighbour upsample (RetinaFace's FPN spells its 2x upsample that way, retina_r50.cpp:156-172:
                // addDeconvolutionNd(256, DimsHW{2, 2}, ones) with setNbGroups(256)).  Same values, exactly; the resize kernel moves
                // 16-byte chunks where the direct transposed convolution walked scalars (253 us per launch on the 1280 x 1280 config).
                {
                    bool ones = l.groups == ti.C && l.nb_out == ti.C && l.kernel[0] == l.stride[0] && l.kernel[1] == l.stride[1] && l.kernel[0] > 1 &&
                                l.padding[0] == 0 && l.padding[1] == 0 && l.dilation[0] == 1 && l.dilation[1] == 1 &&
                                (int64_t)l.w0.size() == (int64_t)ti.C * l.kernel[0] * l.kernel[1];
                    for (size_t i = 0; ones && i < l.w0.size(); ++i) ones = l.w0[i] == 1.0f;
                    for (size_t i = 0; ones && i < l.w1.size(); ++i) ones = l.w1[i] == 0.0f;
                    if (ones) {
                        const int out = new_tensor(l.outputs[0], out_dims(), LAY_NHWC, true);
                        add_op(OP_RESIZE, l.name + " [all-ones depthwise deconvolution = nearest upsample]", {in}, {out});
                        pt_of[l.outputs[0]] = out;
                        return true;
                    }
                }
                const int out = new_tensor(l.outputs[0], out_dims(), LAY_NHWC, true);
                POp& op = add_op(OP_DECONV, l.name, {in}, {out});
                op.src_layer = li;
                ConvArgs& a = op.conv;
                const PTensor& to = plan.tensors[out];
                a.H = ti.H; a.W = ti.W; a.Cin = ti.C; a.Ho = to.H; a.Wo = to.W; a.Cout = to.C;
                a.kh = l.kernel[0]; a.kw = l.kernel[1]; a.stride_h = l.stride[0]; a.stride_w = l.stride[1];
                a.pad_h = l.padding[0]; a.pad_w = l.padding[1]; a.dil_h = l.dilation[0]; a.dil_w = l.dilation[1];
                a.groups = l.groups;
                op.flops = 2.0 * ti.H * ti.W * a.Cin * (double)a.kh * a.kw * (a.Cout / a.groups);
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_POOLING: {
                const int in = need_nhwc(l.inputs[0]);
                const int out = new_tensor(l.outputs[0], out_dims(), LAY_NHWC, true);
                POp& op = add_op(OP_POOL, l.name, {in}, {out});
                op.i[0] = l.op == TRTX_POOLING_MAX ? POOL_MAX : POOL_AVG;
                op.i[1] = l.kernel[0]; op.i[2] = l.kernel[1]; op.i[3] = l.stride[0]; op.i[4] = l.stride[1];
                op.i[5] = l.padding[0]; op.i[6] = l.padding[1]; op.i[7] = l.avg_exclusive;
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_RESIZE: {
                if (l.op != TRTX_RESIZE_NEAREST) return fail(l.name + ": only nearest resize is implemented");
                const Dims& di = net.tensors[l.inputs[0]].dims;
                const Dims& dout = out_dims();
                if (!spatial(di) || dout.d[di.nb - 3] != di.d[di.nb - 3]) return fail(l.name + ": resize must keep channels");
                const int in = need_nhwc(l.inputs[0]);
                const int out = new_tensor(l.outputs[0], dout, LAY_NHWC, true);
                add_op(OP_RESIZE, l.name, {in}, {out});
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_ACTIVATION: {
                const int code = act_code(l.op);
                if (code < 0) return fail(l.name + ": unsupported activation type");
                const int p = pt_of[l.inputs[0]];
                const bool nhwc = plan.tensors[p].layout == LAY_NHWC;
                const int out = new_tensor(l.outputs[0], out_dims(), nhwc ? LAY_NHWC : LAY_LINEAR, plan.tensors[p].batched);
                POp& op = add_op(nhwc ? OP_ACT_NHWC : OP_ACT_LIN, l.name, {p}, {out});
                op.i[0] = code;
                op.f[0] = l.alpha;
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_SCALE: {
                bool pow1 = true;
                for (float p : l.w2) pow1 = pow1 && p == 1.0f;
                const Dims& di = net.tensors[l.inputs[0]].dims;
                if (spatial(di) && pow1 && l.op != TRTX_SCALE_ELEMENTWISE) {
                    const int in = need_nhwc(l.inputs[0]);
                    const int out = new_tensor(l.outputs[0], out_dims(), LAY_NHWC, true);
                    POp& op = add_op(OP_SCALE_NHWC, l.name, {in}, {out});
                    op.src_layer = li;
                    pt_of[l.outputs[0]] = out;
                    return true;
                }
                if (l.op == TRTX_SCALE_ELEMENTWISE) return fail(l.name + ": elementwise scale is not implemented");
                const int in = need_lin(l.inputs[0]);
                const int out = new_tensor(l.outputs[0], out_dims(), LAY_LINEAR, plan.tensors[in].batched);
                POp& op = add_op(OP_SCALE_LIN, l.name, {in}, {out});
                op.src_layer = li;
                op.i[0] = l.op == TRTX_SCALE_CHANNEL ? 1 : 0;
                const int ca = (net.explicit_batch && di.nb >= 4) ? 1 : (di.nb >= 3 ? di.nb - 3 : 0);
                long outer = 1, inner = 1;
                for (int k = 0; k < ca; ++k) outer *= di.d[k];
                for (int k = ca + 1; k < di.nb; ++k) inner *= di.d[k];
                op.i[1] = (int)outer; op.i[2] = (int)di.d[ca]; op.i[3] = (int)inner;
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_ELEMENTWISE: {
                const int code = ew_code(l.op);
                if (code < 0) return fail(l.name + ": unsupported elementwise op");
                const Dims &da = net.tensors[l.inputs[0]].dims, &db = net.tensors[l.inputs[1]].dims;
                const int pa = pt_of[l.inputs[0]], pb = pt_of[l.inputs[1]];
                const bool any_nhwc = plan.tensors[pa].layout == LAY_NHWC || plan.tensors[pb].layout == LAY_NHWC;
                if (spatial(da) && da == db && any_nhwc) {
                    const int a = need_nhwc(l.inputs[0]), b = need_nhwc(l.inputs[1]);
                    const int out = new_tensor(l.outputs[0], out_dims(), LAY_NHWC, true);
                    POp& op = add_op(OP_EW_NHWC, l.name, {a, b}, {out});
                    op.i[0] = code;
                    pt_of[l.outputs[0]] = out;
                    return true;
                }
                const int a = need_lin(l.inputs[0]), b = need_lin(l.inputs[1]);
                const Dims& dout = out_dims();
                const bool batched = plan.tensors[a].batched || plan.tensors[b].batched;
                const int out = new_tensor(l.outputs[0], dout, LAY_LINEAR, batched);
                POp& op = add_op(OP_EW_LIN, l.name, {a, b}, {out});
                op.i[0] = code;
                op.view.rank = dout.nb;
                long sa = 1, sb = 1;
                for (int k = dout.nb - 1; k >= 0; --k) {
                    op.view.shape[k] = dout.d[k];
                    op.view.stride_in[k] = (da.d[k] == 1 && dout.d[k] > 1) ? 0 : sa;
                    op.view.stride_in2[k] = (db.d[k] == 1 && dout.d[k] > 1) ? 0 : sb;
                    sa *= da.d[k];
                    sb *= db.d[k];
                }
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_CONCAT: return emit_concat(li);
            case L_SLICE: return emit_slice(li);
            case L_SHUFFLE: return emit_shuffle(li);
            case L_SOFTMAX: {
                const Dims& di = net.tensors[l.inputs[0]].dims;
                int ax;
                if (l.axis < 0) {
                    ax = std::max(0, di.nb - 3);
                } else {
                    ax = -1;
                    for (int k = 0; k < di.nb; ++k)
                        if ((l.axis >> k) & 1) {
                            if (ax >= 0) return fail(l.name + ": softmax over several axes");
                            ax = k;
                        }
                    if (ax < 0) return fail(l.name + ": softmax without axis");
                }
                const int in = need_lin(l.inputs[0]);
                const int out = new_tensor(l.outputs[0], out_dims(), LAY_LINEAR, plan.tensors[in].batched);
                POp& op = add_op(OP_SOFTMAX, l.name, {in}, {out});
                long outer = 1, inner = 1;
                for (int k = 0; k < ax; ++k) outer *= di.d[k];
                for (int k = ax + 1; k < di.nb; ++k) inner *= di.d[k];
                op.i[0] = (int)outer; op.i[1] = (int)di.d[ax]; op.i[2] = (int)inner;
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_MATMUL: {
                const Dims &da = net.tensors[l.inputs[0]].dims, &db = net.tensors[l.inputs[1]].dims;
                for (int k = 0; k < da.nb - 2; ++k)
                    if (da.d[k] != 1 || db.d[k] != 1) return fail(l.name + ": matmul with leading dims > 1 is not implemented");
                if (l.mm_op[0] == TRTX_MATMUL_VECTOR || l.mm_op[1] == TRTX_MATMUL_VECTOR) return fail(l.name + ": kVECTOR matmul");
                const int a = need_lin(l.inputs[0]), b = need_lin(l.inputs[1]);
                const Dims& dout = out_dims();
                const int out = new_tensor(l.outputs[0], dout, LAY_LINEAR, plan.tensors[a].batched || plan.tensors[b].batched);
                POp& op = add_op(OP_MATMUL, l.name, {a, b}, {out});
                const int n = da.nb;
                const bool ta = l.mm_op[0] == TRTX_MATMUL_TRANSPOSE, tb = l.mm_op[1] == TRTX_MATMUL_TRANSPOSE;
                op.i[0] = (int)dout.d[n - 2]; op.i[1] = (int)dout.d[n - 1];
                op.i[2] = (int)(ta ? da.d[n - 2] : da.d[n - 1]); op.i[3] = ta; op.i[4] = tb;
                op.flops = 2.0 * op.i[0] * op.i[1] * op.i[2];
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_CONSTANT: {
                const int out = new_tensor(l.outputs[0], out_dims(), LAY_LINEAR, false);
                plan.tensors[out].batched = false;
                Storage s;
                s.kind = ST_WEIGHTS;
                s.bytes = (size_t)out_dims().volume() * 4;
                plan.tensors[out].storage = (int)plan.storages.size();
                plan.storages.push_back(s);
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_REDUCE: {
                const Dims& di = net.tensors[l.inputs[0]].dims;
                const int p = pt_of[l.inputs[0]];
                const int hw_mask = 0b110 << (di.nb - 3);
                if (spatial(di) && plan.tensors[p].layout == LAY_NHWC && l.op == TRTX_REDUCE_AVG && l.axis == hw_mask && l.keep_dims) {
                    const int out = new_tensor(l.outputs[0], out_dims(), LAY_NHWC, true);
                    add_op(OP_REDUCE_HW, l.name, {p}, {out});
                    pt_of[l.outputs[0]] = out;
                    return true;
                }
                // contiguous run of reduced axes in LINEAR layout
                int first = -1, last = -1;
                for (int k = 0; k < di.nb; ++k)
                    if ((l.axis >> k) & 1) {
                        if (first < 0) first = k;
                        if (last >= 0 && k != last + 1) return fail(l.name + ": non-contiguous reduce axes");
                        last = k;
                    }
                if (first < 0) return fail(l.name + ": reduce without axes");
                int rop;
                switch (l.op) {
                    case TRTX_REDUCE_SUM: rop = 0; break;
                    case TRTX_REDUCE_AVG: rop = 1; break;
                    case TRTX_REDUCE_MAX: rop = 2; break;
                    default: return fail(l.name + ": unsupported reduce op");
                }
                const int in = need_lin(l.inputs[0]);
                const int out = new_tensor(l.outputs[0], out_dims(), LAY_LINEAR, plan.tensors[in].batched);
                POp& op = add_op(OP_REDUCE_LIN, l.name, {in}, {out});
                long outer = 1, axis = 1, inner = 1;
                for (int k = 0; k < first; ++k) outer *= di.d[k];
                for (int k = first; k <= last; ++k) axis *= di.d[k];
                for (int k = last + 1; k < di.nb; ++k) inner *= di.d[k];
                op.i[0] = rop; op.i[1] = (int)outer; op.i[2] = (int)axis; op.i[3] = (int)inner;
                pt_of[l.outputs[0]] = out;
                return true;
            }
            case L_PLUGIN: {
                // "RoiAlign" (rcnn/RoiAlignPlugin.h; blob int res, float scale, int ratio, int nProp, int C, int fh, int fw) in an fp16
                // engine whose feature map already lives in NHWC fp16: run the engine-native kernel on it and emit the NHWC
                // [P][res][res][C] tensor the res5 convolutions read, instead of fp32 LINEAR in / out plus two layout passes
                // (803 MB fp32 written, re-read and re-written as fp16 per image at C5).  Same detectron2 ROIAlign(aligned=True)
                // arithmetic as the plugin (pinned on the reference's kernel in tests/test_ref_pinning.py); fp32 engines and
                // TRTX_ROIALIGN_PLUGIN=1 keep the plugin route.
                const bool keep_plugin = !opt.roialign_fused;
                if (!keep_plugin && dt == DT_F16 && !net.explicit_batch && l.plugin && l.plugin->type() == "RoiAlign" && l.plugin->version() == "1" &&
                    l.inputs.size() == 2 && l.outputs.size() == 1 && plan.tensors[pt_of[l.inputs[1]]].layout == LAY_NHWC) {
                    const std::vector<uint8_t> blob = l.plugin->serialize();
                    const PTensor& feat = plan.tensors[pt_of[l.inputs[1]]];
                    int32_t iv[7];
                    if (blob.size() == 28) {
                        memcpy(iv, blob.data(), 28);
                        float scale;
                        memcpy(&scale, blob.data() + 4, 4);
                        const Dims& od = net.tensors[l.outputs[0]].dims;
                        if (iv[0] > 0 && iv[3] > 0 && iv[4] == feat.C && iv[5] == feat.H && iv[6] == feat.W && feat.nmul == 1 && feat.C % 8 == 0 &&
                            od.nb == 4 && od.d[0] == iv[3] && od.d[1] == iv[4] && od.d[2] == iv[0] && od.d[3] == iv[0]) {
                            const int boxes = need_lin(l.inputs[0]);
                            // Every reader a 1x1 stride-2 unpadded convolution (res5.0's conv1 and its shortcut, rcnn/backbone.hpp:9,110-117
                            // STRIDE_IN_1X1; Faster R-CNN R50-C4): only the even bins are ever read.  Emit exactly those - [P][7][7][C]
                            // instead of [P][14][14][C] - and run the readers at stride 1 over it: the same samples in the same order
                            // (bit-identical), a quarter of the RoIAlign work and writes (C5 b4: 1.6 GB -> 0.4 GB per step), and the
                            // convolutions read contiguous pixels.  Mask R-CNN's mask-head RoIAlign feeds a stride-1 reader and keeps
                            // the full grid (rcnn/rcnn.cpp:204-233).  TRTX_ROIALIGN_FOLD_STRIDE=0 keeps the full grid (A/B, tests).
                            int step = 0;
                            const bool no_fold = !opt.roialign_fold_stride;
                            if (!no_fold && !net.tensors[l.outputs[0]].is_output && !consumers[l.outputs[0]].empty()) {
                                step = -1;
                                for (int c : consumers[l.outputs[0]]) {
                                    const LayerDef& cl = net.layers[c];
                                    const bool ok = cl.kind == L_CONV && cl.inputs[0] == l.outputs[0] && cl.kernel[0] == 1 && cl.kernel[1] == 1 &&
                                                    cl.stride[0] == cl.stride[1] && cl.stride[0] > 1 && cl.padding[0] == 0 && cl.padding[1] == 0 &&
                                                    cl.groups == 1 && (step < 0 || step == cl.stride[0]);
                                    if (!ok) { step = 0; break; }
                                    step = cl.stride[0];
                                }
                            }
                            Dims od_emit = od;
                            if (step > 1) {
                                od_emit.d[2] = od_emit.d[3] = (iv[0] - 1) / step + 1;
                                for (int c : consumers[l.outputs[0]]) stride_folded[c] = 1;
                            } else {
                                step = 1;
                            }
                            const int out = new_tensor(l.outputs[0], od_emit, LAY_NHWC, true);
                            POp& op = add_op(OP_ROI_ALIGN, l.name + (step > 1 ? " [native NHWC, every 2nd bin]" : " [native NHWC]"), {boxes, pt_of[l.inputs[1]]}, {out});
                            op.i[0] = iv[0]; op.i[1] = iv[2]; op.i[2] = iv[3]; op.i[3] = step;
                            op.f[0] = scale;
                            op.bytes = 2.0 * (double)od_emit.volume() + 2.0 * (double)feat.C * feat.H * feat.W;
                            pt_of[l.outputs[0]] = out;
                            return true;
                        }
                    }
                }
                if (is_builtin_mish(li)) {  // a Mish_TRT no convolution absorbed: an activation op in the layout its input already has
                    const int p = pt_of[l.inputs[0]];
                    const bool nhwc = plan.tensors[p].layout == LAY_NHWC;
                    const int out = new_tensor(l.outputs[0], net.tensors[l.outputs[0]].dims, nhwc ? LAY_NHWC : LAY_LINEAR, plan.tensors[p].batched);
                    POp& op = add_op(nhwc ? OP_ACT_NHWC : OP_ACT_LIN, l.name, {p}, {out});
                    op.i[0] = ACT_MISH;
                    op.f[0] = 0.f;
                    pt_of[l.outputs[0]] = out;
                    return true;
                }
                std::vector<int> ins, outs;
                for (int t : l.inputs) ins.push_back(need_lin(t));
                for (size_t s = 0; s < l.outputs.size(); ++s) {
                    const int o = new_tensor(l.outputs[s], net.tensors[l.outputs[s]].dims, LAY_LINEAR, true);
                    outs.push_back(o);
                    pt_of[l.outputs[s]] = o;
                }
                POp& op = add_op(OP_PLUGIN, l.name, ins, outs);
                op.plugin = l.plugin;
                return true;
            }
            case L_IDENTITY: {
                const int p = pt_of[l.inputs[0]];
                pt_of[l.outputs[0]] = plan.tensors[p].layout == LAY_NHWC
                                              ? new_view_nhwc(p, 0, plan.tensors[p].C, l.outputs[0])
                                              : new_view_lin(p, out_dims(), l.outputs[0]);
                return true;
            }
            default: return fail(l.name + ": unknown layer kind");
        }
    }

    bool emit_concat(int li) {
        const LayerDef& l = net.layers[li];
        const Dims& dout = net.tensors[l.outputs[0]].dims;
        const int cax = dout.nb - 3;
        bool all_nhwc = spatial(dout) && l.axis == cax;
        for (int t : l.inputs) all_nhwc = all_nhwc && plan.tensors[pt_of[t]].layout == LAY_NHWC;
        if (all_nhwc) {
            std::vector<int> ins;
            for (int t : l.inputs) ins.push_back(pt_of[t]);
            // concat of consecutive channel views of one tensor == a view of that tensor (C2F, block.cpp:134-141)
            bool consecutive = plan.tensors[ins[0]].parent >= 0;
            for (size_t k = 1; consecutive && k < ins.size(); ++k) {
                const PTensor &a = plan.tensors[ins[k - 1]], &b = plan.tensors[ins[k]];
                consecutive = b.parent == a.parent && b.coff == a.coff + a.C;
            }
            if (consecutive) {
                pt_of[l.outputs[0]] = new_view_nhwc(plan.tensors[ins[0]].parent, plan.tensors[ins[0]].coff,
                                                    (int)dout.d[cax], l.outputs[0]);
                return true;
            }
            const int out = new_tensor(l.outputs[0], dout, LAY_NHWC, true);
            int off = 0;
            for (size_t k = 0; k < ins.size(); ++k) {
                const int c = plan.tensors[ins[k]].C;
                if (!try_place(ins[k], out, off)) {
                    const int v = new_view_nhwc(out, off, c, -1);
                    add_op(OP_COPY_NHWC, l.name + ":copy" + std::to_string(k), {ins[k]}, {v});
                }
                off += c;
            }
            pt_of[l.outputs[0]] = out;
            return true;
        }
        // LINEAR: scatter every input into the dense output
        std::vector<int> ins;
        bool batched = false;
        for (int t : l.inputs) {
            ins.push_back(need_lin(t));
            batched = batched || plan.tensors[ins.back()].batched;
        }
        const int out = new_tensor(l.outputs[0], dout, LAY_LINEAR, batched);
        long ostride[8];
        long s = 1;
        for (int k = dout.nb - 1; k >= 0; --k) {
            ostride[k] = s;
            s *= dout.d[k];
        }
        long pos = 0;
        for (size_t k = 0; k < ins.size(); ++k) {
            const Dims& di = net.tensors[l.inputs[k]].dims;
            POp& op = add_op(OP_SCATTER, l.name + ":in" + std::to_string(k), {ins[k]}, {out});
            op.view.rank = di.nb;
            for (int d = 0; d < di.nb; ++d) {
                op.view.shape[d] = di.d[d];
                op.view.stride_in[d] = ostride[d];
            }
            op.off0 = pos * ostride[l.axis];
            pos += di.d[l.axis];
        }
        pt_of[l.outputs[0]] = out;
        return true;
    }

    bool emit_slice(int li) {
        const LayerDef& l = net.layers[li];
        const Dims& di = net.tensors[l.inputs[0]].dims;
        const int p = pt_of[l.inputs[0]];
        const int cax = di.nb - 3;
        if (spatial(di) && plan.tensors[p].layout == LAY_NHWC) {
            bool chan_only = l.step.d[cax] == 1;
            for (int k = 0; k < di.nb; ++k)
                if (k != cax) chan_only = chan_only && l.start.d[k] == 0 && l.size.d[k] == di.d[k] && l.step.d[k] == 1;
            if (chan_only) {
                pt_of[l.outputs[0]] = new_view_nhwc(p, (int)l.start.d[cax], (int)l.size.d[cax], l.outputs[0]);
                return true;
            }
        }
        const int in = need_lin(l.inputs[0]);
        const int out = new_tensor(l.outputs[0], l.size, LAY_LINEAR, plan.tensors[in].batched);
        POp& op = add_op(OP_GATHER, l.name, {in}, {out});
        op.view.rank = di.nb;
        long s = 1, off = 0;
        for (int k = di.nb - 1; k >= 0; --k) {
            op.view.shape[k] = l.size.d[k];
            op.view.stride_in[k] = s * l.step.d[k];
            off += l.start.d[k] * s;
            s *= di.d[k];
        }
        op.off0 = off;
        pt_of[l.outputs[0]] = out;
        return true;
    }

    bool emit_shuffle(int li) {
        const LayerDef& l = net.layers[li];
        const Dims& di = net.tensors[l.inputs[0]].dims;
        const Dims& dout = net.tensors[l.outputs[0]].dims;
        auto identity = [](const int32_t* p, int n) {
            for (int k = 0; k < n; ++k)
                if (p[k] != k) return false;
            return true;
        };
        int cur = need_lin(l.inputs[0]);
        Dims dcur = di;
        if (!identity(l.perm1, di.nb)) {
            Dims t = di;
            for (int k = 0; k < di.nb; ++k) t.d[k] = di.d[l.perm1[k]];
            const int q = new_tensor(-1, t, LAY_LINEAR, plan.tensors[cur].batched);
            POp& op = add_op(OP_GATHER, l.name + ":t1", {cur}, {q});
            long st[8], s = 1;
            for (int k = di.nb - 1; k >= 0; --k) {
                st[k] = s;
                s *= di.d[k];
            }
            op.view.rank = di.nb;
            for (int k = 0; k < di.nb; ++k) {
                op.view.shape[k] = t.d[k];
                op.view.stride_in[k] = st[l.perm1[k]];
            }
            cur = q;
            dcur = t;
        }
        // reshape: a view
        Dims r = dcur;
        if (l.reshape.nb > 0) {
            // the network already validated/inferred the reshape; recover it from the output dims
            r.nb = dout.nb;
            for (int k = 0; k < dout.nb; ++k) r.d[k] = 0;
            // invert perm2: out[k] = r[perm2[k]]
            for (int k = 0; k < dout.nb; ++k) r.d[l.perm2[k]] = dout.d[k];
        }
        if (identity(l.perm2, r.nb)) {
            pt_of[l.outputs[0]] = new_view_lin(cur, dout, l.outputs[0]);
            return true;
        }
        const int rv = new_view_lin(cur, r, -1);
        const int out = new_tensor(l.outputs[0], dout, LAY_LINEAR, plan.tensors[cur].batched);
        POp& op = add_op(OP_GATHER, l.name + ":t2", {rv}, {out});
        long st[8], s = 1;
        for (int k = r.nb - 1; k >= 0; --k) {
            st[k] = s;
            s *= r.d[k];
        }
        op.view.rank = r.nb;
        for (int k = 0; k < r.nb; ++k) {
            op.view.shape[k] = dout.d[k];
            op.view.stride_in[k] = st[l.perm2[k]];
        }
        pt_of[l.outputs[0]] = out;
        return true;
    }

    void apply_aliases() {
        for (auto& a : aliases)
            if (pt_of[a.first] < 0 && pt_of[a.second] >= 0) pt_of[a.first] = pt_of[a.second];
    }

    // ---- driver ----------------------------------------------------------------------------------------
    bool run() {
        plan.explicit_batch = net.explicit_batch;
        plan.fp16 = net.fp16;
        plan.max_batch = net.explicit_batch ? 1 : net.max_batch;
        // input bindings
        for (int t : net.input_ids()) {
            const int p = new_tensor(t, net.tensors[t].dims, LAY_LINEAR, true);
            Storage s;
            s.kind = ST_BINDING;
            s.binding = (int)plan.binding_tensor.size();
            plan.tensors[p].storage = (int)plan.storages.size();
            plan.storages.push_back(s);
            plan.binding_tensor.push_back(t);
            plan.binding_ptensor.push_back(p);
            plan.binding_is_input.push_back(true);
            pt_of[t] = p;
        }
        analyse_yolo_head();  // before conv fusion: it claims the DFL 1x1 convolutions
        analyse_fusion();
        for (size_t li = 0; li < net.layers.size(); ++li) {
            if (yolo_at[li] >= 0) {
                if (!emit_yolo_head(yolo_heads[yolo_at[li]])) return false;
                continue;
            }
            if (group_at[li] >= 0) {
                if (!emit_conv(groups[group_at[li]])) return false;
                continue;
            }
            if (absorbed[li]) {
                apply_aliases();
                continue;
            }
            if (!emit_layer((int)li)) return false;
            apply_aliases();
        }
        // output bindings: LINEAR fp32
        for (int t : net.output_ids()) {
            if (pt_of[t] < 0) return fail("output tensor " + net.tensors[t].name + " is never produced");
            int p = need_lin(t);
            PTensor& pt = plan.tensors[p];
            const bool own = pt.parent < 0 && pt.storage < 0 && !is_binding_tensor(p);
            if (!own) {
                const int q = new_tensor(t, pt.dims, LAY_LINEAR, pt.batched);
                POp& op = add_op(OP_COPY_LIN, "output:" + net.tensors[t].name, {p}, {q});
                op.i[0] = 0;
                p = q;
            }
            Storage s;
            s.kind = ST_BINDING;
            s.binding = (int)plan.binding_tensor.size();
            plan.tensors[p].storage = (int)plan.storages.size();
            plan.storages.push_back(s);
            plan.binding_tensor.push_back(t);
            plan.binding_ptensor.push_back(p);
            plan.binding_is_input.push_back(false);
        }
        return finalize();
    }

    size_t tensor_bytes(const PTensor& t) const {
        const size_t es = dtype_size(t.dtype);
        if (t.layout == LAY_NHWC) {
            const size_t n = (t.nfix ? (size_t)t.nfix : (size_t)plan.max_batch) * (size_t)t.nmul;
            return n * t.H * t.W * (size_t)t.Calloc * es;
        }
        return (t.batched ? (size_t)plan.max_batch : 1) * (size_t)t.dims.volume() * es;
    }

    // kINT8: which NHWC tensors live in int8.  An owning tensor (a conv output, or a concat buffer several producers write
    // slices of) becomes int8 when it has a calibrated scale and EVERY op touching it can work on int8 in place: MFMA-eligible
    // convolutions (as producer, consumer or residual) and nearest resizes.  Anything else (pool chains, the fused detect
    // head, layout conversions, plugins, depth-to-space ...) keeps the tensor in fp16, and a convolution simply dequantises /
    // requantises at that boundary in its epilogue.  The scale is a property of the OWNER, so all producers of a concat
    // buffer quantise to the same scale (TensorRT reaches the same end by forcing equal scales on concat inputs).
    // `veto`: owners that an earlier attempt put in int8 and finalize's kernel choice then could not serve (see finalize()).
    void assign_int8(const std::vector<char>& veto) {
        if (!net.int8 || dt != DT_F16) return;
        const int nt = (int)plan.tensors.size();
        auto top = [&](int t) {
            while (plan.tensors[t].parent >= 0) t = plan.tensors[t].parent;
            return t;
        };
        std::vector<char> cand(nt, 0);
        for (const PTensor& t : plan.tensors)
            if (t.parent < 0 && t.layout == LAY_NHWC && t.dtype == DT_F16 && t.net_tensor >= 0 && t.net_tensor < (int)net.tensor_scale.size() &&
                net.tensor_scale[t.net_tensor] > 0.f && t.C % 16 == 0 && t.nmul == 1 && !is_binding_tensor(t.id) && !veto[t.id])
                cand[t.id] = 1;
        auto view_ok = [&](int t) {
            int off = 0;
            owner_of(t, &off);
            return off % 16 == 0 && plan.tensors[t].C % 16 == 0;
        };
        auto conv_ok = [&](const POp& op) {
            const ConvArgs& a = op.conv;
            return op.kind == OP_CONV && !op.stem && !op.from_deconv && a.groups == 1 && a.dil_h == 1 && a.dil_w == 1 && a.kh * a.kw <= 30;
        };
        for (const POp& op : plan.ops) {
            for (size_t j = 0; j < op.in.size(); ++j) {
                const PTensor& t = plan.tensors[op.in[j]];
                if (t.layout != LAY_NHWC) continue;
                bool ok = false;
                if (conv_ok(op)) ok = view_ok(op.in[j]) && (j > 0 || op.conv.kh * op.conv.kw * t.C >= 32);
                else if (op.kind == OP_RESIZE) ok = view_ok(op.in[j]);
                if (!ok) cand[top(op.in[j])] = 0;
            }
            for (int o : op.out) {
                if (plan.tensors[o].layout != LAY_NHWC) continue;
                const bool ok = (conv_ok(op) || op.kind == OP_RESIZE) && view_ok(o);
                if (!ok) cand[top(o)] = 0;
            }
        }
        for (bool changed = true; changed;) {
            changed = false;
            auto tie = [&](int a, int b) {  // both int8 or neither
                if (cand[a] != cand[b]) {
                    cand[a] = cand[b] = 0;
                    changed = true;
                }
            };
            for (const POp& op : plan.ops) {
                if (op.kind == OP_CONV && op.in.size() > 1) tie(top(op.in[1]), top(op.out[0]));
                if (op.kind == OP_RESIZE) tie(top(op.in[0]), top(op.out[0]));
            }
        }
        for (PTensor& t : plan.tensors) {
            if (t.layout != LAY_NHWC) continue;
            const int o = top(t.id);
            if (!cand[o]) continue;
            t.dtype = DT_I8;
            t.scale = net.tensor_scale[plan.tensors[o].net_tensor];
        }
    }

    // Independent convolutions of one kernel instantiation -> one launch (OP_CONV_GROUP; kernels/conv_igemm.hip conv_igemm_group_f16_kernel).
    // The YOLOv8 detect head is six chains of depth three over three pyramid levels (yolov8/src/model.cpp:188-251): cv2.{0,1,2}.0 are three
    // independent 3x3 convolutions to 64 channels, cv3.{0,1,2}.0 three to 80, and so on down the chains - 18 launches, of which the 20x20 and
    // 40x40 levels (100 / 400 tiles at batch 32 for 256 CUs) mostly pay the per-launch floor.  Members must be pairwise independent (no
    // dependency path either way), have the same filter, stride, Cout, activation and residual-ness and sit at the same height above the
    // plan's sinks (so that they are the same LAYER of sibling branches, not unrelated work that happens to fit) and pass conv_igemm_group_supported().  The ops are then re-ordered
    // (a topological order of the dependency graph with each group contracted to one node; a grouping that would close a cycle between
    // two groups is dropped) and every group's members are replaced by one op whose in / out are the unions.  Each member is computed
    // exactly as its own launch computes it: bit-identical outputs.  TRTX_GROUP_CONVS=0 keeps one launch per convolution (A/B, tests).
    void group_convs() {
        if (dt != DT_F16 || CalibrationLowering::active()) return;   // (INT8 plans group too since round 5: members of one storage mix - same in / out / shortcut int8 flags)
        bool mark_only = false;   // TRTX_GROUP_CONVS=0: one launch per convolution, but the would-be members keep the group's K order (t_wsk = 1): same bits
        if (!opt.group_convs) mark_only = true;
        const int n = (int)plan.ops.size();
        if (n < 2 || n > 4096) return;
        // all dependencies (RAW, WAR, WAW at storage / channel-range granularity, as finalize step 5 computes them) in the current order
        struct Acc { int storage; long lo, hi; int op; bool write; };
        auto acc_of = [&](int t, int op, bool write) {
            const PTensor& pt = plan.tensors[t];
            Acc a{pt.storage, 0, 0, op, write};
            if (pt.layout == LAY_NHWC) {
                a.lo = pt.rcoff;
                a.hi = pt.rcoff + (pt.parent < 0 && pt.Calloc > pt.C ? pt.Calloc : pt.C);
            } else {
                a.lo = pt.reoff;
                a.hi = pt.reoff + pt.dims.volume();
            }
            return a;
        };
        std::vector<std::vector<int>> deps(n);
        {
            std::vector<Acc> log;
            for (int k = 0; k < n; ++k) {
                const POp& op = plan.ops[k];
                std::vector<Acc> mine;
                for (int t : op.in) mine.push_back(acc_of(t, k, false));
                for (int t : op.extra_in) mine.push_back(acc_of(t, k, false));
                for (int t : op.out) mine.push_back(acc_of(t, k, true));
                for (const Acc& m : mine)
                    for (const Acc& o : log)
                        if (o.storage == m.storage && o.lo < m.hi && m.lo < o.hi && (o.write || m.write) && o.op != k) deps[k].push_back(o.op);
                std::sort(deps[k].begin(), deps[k].end());
                deps[k].erase(std::unique(deps[k].begin(), deps[k].end()), deps[k].end());
                log.insert(log.end(), mine.begin(), mine.end());
            }
        }
        // ancestors (transitive), as bit rows
        const int words = (n + 63) / 64;
        std::vector<uint64_t> anc((size_t)n * words, 0);
        auto is_anc = [&](int a, int of) { return (anc[(size_t)of * words + a / 64] >> (a % 64)) & 1ull; };
        for (int k = 0; k < n; ++k)
            for (int d : deps[k]) {
                anc[(size_t)k * words + d / 64] |= 1ull << (d % 64);
                for (int w = 0; w < words; ++w) anc[(size_t)k * words + w] |= anc[(size_t)d * words + w];
            }
        auto args_at_max_batch = [&](const POp& op) {
            ConvArgs a = op.conv;
            const PTensor& ti = plan.tensors[op.in[0]];
            a.N = (ti.nfix ? ti.nfix : plan.max_batch) * ti.nmul;
            a.M = a.N * a.Ho * a.Wo;
            a.residual = op.in.size() > 1 ? reinterpret_cast<const void*>(1) : nullptr;   // presence only (alignment rules look at ld_res)
            return a;
        };
        auto candidate = [&](const POp& op) {
            if (op.kind != OP_CONV || !op.igemm || op.stem || op.from_deconv || !op.extra_in.empty()) return false;
            ConvArgs two[2] = {args_at_max_batch(op), args_at_max_batch(op)};
            return conv_igemm_group_supported(two, 2);
        };
        auto same_layer_shape = [&](const POp& x, const POp& y) {
            const ConvArgs &a = x.conv, &b = y.conv;
            return a.kh == b.kh && a.kw == b.kw && a.stride_h == b.stride_h && a.stride_w == b.stride_w && a.pad_h == b.pad_h && a.pad_w == b.pad_w &&
                   a.Cout == b.Cout && a.act1 == b.act1 && a.act2 == b.act2 && a.alpha1 == b.alpha1 && a.alpha2 == b.alpha2 && x.in.size() == y.in.size() &&
                   a.in_i8 == b.in_i8 && a.out_i8 == b.out_i8 && a.res_i8 == b.res_i8;
        };
        // height = longest dependency path from an op down to a sink: sibling branches that end in the same consumer (the three levels'
        // arms into the fused head op) put their corresponding layers at equal heights; a bottleneck of the neck that merely has the same
        // shape as an arm's convolution sits higher and stays out of the arm's group
        std::vector<int> height(n, 0);
        for (int k = n - 1; k >= 0; --k)
            for (int d : deps[k]) height[d] = std::max(height[d], height[k] + 1);
        std::vector<int> group_of(n, -1);
        std::vector<std::vector<int>> groups;
        auto acyclic = [&]() {   // the dependency graph with every group contracted to one node
            std::vector<int> node(n);
            int nn = 0;
            std::vector<int> gnode(groups.size(), -1);
            for (int k = 0; k < n; ++k) {
                if (group_of[k] >= 0) {
                    if (gnode[group_of[k]] < 0) gnode[group_of[k]] = nn++;
                    node[k] = gnode[group_of[k]];
                } else {
                    node[k] = nn++;
                }
            }
            std::vector<std::vector<int>> succ(nn);
            std::vector<int> indeg(nn, 0);
            for (int k = 0; k < n; ++k)
                for (int d : deps[k])
                    if (node[d] != node[k]) {
                        succ[node[d]].push_back(node[k]);
                        ++indeg[node[k]];
                    }
            std::vector<int> q;
            for (int v = 0; v < nn; ++v)
                if (!indeg[v]) q.push_back(v);
            size_t done = 0;
            while (done < q.size()) {
                const int v = q[done++];
                for (int w : succ[v])
                    if (--indeg[w] == 0) q.push_back(w);
            }
            return (int)q.size() == nn;
        };
        for (int k = 0; k < n; ++k) {
            if (group_of[k] >= 0 || !candidate(plan.ops[k])) continue;
            std::vector<int> mem = {k};
            std::vector<ConvArgs> margs = {args_at_max_batch(plan.ops[k])};
            for (int j = k + 1; j < n && (int)mem.size() < kMaxConvGroup; ++j) {
                if (group_of[j] >= 0 || height[j] != height[k] || !candidate(plan.ops[j]) || !same_layer_shape(plan.ops[k], plan.ops[j])) continue;
                bool indep = true;
                for (int m : mem) indep = indep && !is_anc(m, j) && !is_anc(j, m);
                if (!indep) continue;
                margs.push_back(args_at_max_batch(plan.ops[j]));
                // the operand path (t_rs: registers or LDS-DMA, the same bits either way) is a per-layer heuristic; a group runs on the
                // path of its member with the most rows.  Checked on a COPY (a rejected candidate used to leave its t_rs on the others)
                std::vector<ConvArgs> trial = margs;
                int big = 0;
                for (size_t q = 1; q < trial.size(); ++q)
                    if (trial[q].M > trial[big].M) big = (int)q;
                const int rs = trial[big].t_rs;
                for (ConvArgs& ma : trial) ma.t_rs = rs;
                if (!conv_igemm_group_supported(trial.data(), (int)trial.size())) {
                    margs.pop_back();
                    continue;
                }
                mem.push_back(j);
            }
            if (mem.size() < 2) continue;
            const int gi = (int)groups.size();
            groups.push_back(mem);
            for (int m : mem) group_of[m] = gi;
            if (!acyclic()) {   // (two groups each waiting for a member of the other): leave these convolutions alone
                for (int m : mem) group_of[m] = -1;
                groups.pop_back();
            }
        }
        if (groups.empty()) return;
        if (mark_only) {
            for (const auto& mem : groups)
                for (int m : mem) {
                    plan.ops[m].conv.t_wsk = 1;
                    plan.ops[m].conv.k_pinned = 1;   // ... and the tuner keeps it that way (its candidates for a pinned layer all walk K as the main kernel does)
                }
            return;
        }
        // new order: Kahn over the contracted graph, ready nodes taken in the order of their first member's old position
        std::vector<int> node(n);
        int nn = 0;
        std::vector<int> gnode(groups.size(), -1), first_op;
        for (int k = 0; k < n; ++k) {
            if (group_of[k] >= 0 && gnode[group_of[k]] >= 0) {
                node[k] = gnode[group_of[k]];
                continue;
            }
            if (group_of[k] >= 0) gnode[group_of[k]] = nn;
            node[k] = nn++;
            first_op.push_back(k);
        }
        std::vector<std::vector<int>> succ(nn);
        std::vector<int> indeg(nn, 0);
        for (int k = 0; k < n; ++k)
            for (int d : deps[k])
                if (node[d] != node[k]) {
                    succ[node[d]].push_back(node[k]);
                    ++indeg[node[k]];
                }
        std::vector<char> emitted(nn, 0);
        std::vector<POp> out;
        out.reserve(nn);
        for (int step = 0; step < nn; ++step) {
            int pick = -1;
            for (int v = 0; v < nn; ++v)
                if (!emitted[v] && indeg[v] == 0) { pick = v; break; }   // nodes are numbered by first member: the lowest ready one
            if (pick < 0) return;   // cannot happen (acyclic() held); keep the plan as it was
            emitted[pick] = 1;
            for (int w : succ[pick]) --indeg[w];
            const int k0 = first_op[pick];
            if (group_of[k0] < 0) {
                out.push_back(plan.ops[k0]);
                continue;
            }
            POp g;
            g.kind = OP_CONV_GROUP;
            g.dtype = plan.ops[k0].dtype;
            g.conv = plan.ops[k0].conv;
            int big = groups[group_of[k0]][0];
            for (int m : groups[group_of[k0]])
                if ((long)plan.ops[m].conv.Ho * plan.ops[m].conv.Wo * plan.tensors[plan.ops[m].in[0]].nmul >
                    (long)plan.ops[big].conv.Ho * plan.ops[big].conv.Wo * plan.tensors[plan.ops[big].in[0]].nmul)
                    big = m;
            for (int m : groups[group_of[k0]]) {
                POp mo = plan.ops[m];
                mo.conv.t_rs = plan.ops[big].conv.t_rs;
                // ONE summation order per member, grouped or not: the grouped kernel walks K as the main kernel does, so the member's own launch -
                // the fallback the executor takes at a batch where the members no longer share an instantiation, and TRTX_GROUP_CONVS=0 - must not pick
                // the wave-split-K variant (different K order: the same engine rounded differently depending on batch size and on the switch)
                mo.conv.t_wsk = 1;
                mo.conv.k_pinned = 1;
                g.group.push_back(mo);
                g.name += (g.name.empty() ? "" : " + ") + mo.name;
                for (int t : mo.in)
                    if (std::find(g.in.begin(), g.in.end(), t) == g.in.end()) g.in.push_back(t);
                g.out.push_back(mo.out[0]);
                g.flops += mo.flops;
                g.bytes += mo.bytes;
            }
            g.name = "[group of " + std::to_string(g.group.size()) + "] " + g.name;
            out.push_back(std::move(g));
        }
        plan.ops = std::move(out);
    }

    // Upsample -> Concat -> Conv1x1 without the upsampled tensor (YOLOv8 head: model.cpp:130-160, twice per network).  The nearest 2x
    // resize writes the first channel slice of a concat buffer whose only reader is a 1x1 stride-1 convolution: that convolution's
    // A-gather can fetch those channels from the half-resolution tensor at (h >> 1, w >> 1) itself (ConvArgs::up_in).  The resize launch,
    // its write of the 4x larger tensor and the convolution's read of it disappear (YOLOv8n b32: 2 launches, 78 MB written + 78 MB read
    // per step become 19.5 MB read); every product is formed from the same operands in the same order: bit-identical outputs.
    // TRTX_FOLD_UPSAMPLE=0 keeps the resize (A/B, tests).  Not with kINT8 (the int8 resize requantises between two scales).
    void fold_upsample() {
        if (net.int8 || CalibrationLowering::active()) return;   // (fp16 and fp32 engines: both MFMA kernels fetch the slice from the half-resolution tensor)
        if (!opt.fold_upsample) return;
        auto top = [&](int t) {
            while (plan.tensors[t].parent >= 0) t = plan.tensors[t].parent;
            return t;
        };
        for (size_t k = 0; k < plan.ops.size(); ++k) {
            const POp& rz = plan.ops[k];
            if (rz.kind != OP_RESIZE) continue;
            const PTensor& src = plan.tensors[rz.in[0]];
            const PTensor& up = plan.tensors[rz.out[0]];
            if (up.parent < 0 || up.rcoff != 0 || up.layout != LAY_NHWC || src.layout != LAY_NHWC || up.H != 2 * src.H || up.W != 2 * src.W || up.C != src.C ||
                up.C % 64 || src.ld % 8 || src.rcoff % 8 || src.nmul != 1 || up.nmul != 1 || is_binding_tensor(rz.out[0]))
                continue;
            const int owner = top(rz.out[0]);
            // every reader of the buffer: exactly one, a 1x1 stride-1 convolution over the WHOLE buffer; nobody reads the slice itself
            int reader = -1, n_readers = 0;
            for (size_t j = 0; j < plan.ops.size(); ++j)
                for (int t : plan.ops[j].in)
                    if (top(t) == owner) {
                        // readers of OTHER channel ranges of the buffer (a skip connection also feeding elsewhere) do not matter
                        const PTensor& rt = plan.tensors[t];
                        if (rt.rcoff < up.C) {
                            reader = (int)j;
                            ++n_readers;
                        }
                    }
            if (n_readers != 1 || is_binding_tensor(owner)) continue;
            POp& cv = plan.ops[reader];
            const ConvArgs& a = cv.conv;
            if (cv.kind != OP_CONV || cv.stem || cv.from_deconv || cv.in[0] != owner || !cv.extra_in.empty() || a.kh != 1 || a.kw != 1 || a.stride_h != 1 ||
                a.stride_w != 1 || a.pad_h || a.pad_w || a.groups != 1 || a.Cin != plan.tensors[owner].C || up.C >= a.Cin || reader < (int)k)
                continue;
            POp probe = cv;
            probe.extra_in = {rz.in[0]};
            if (!choose_conv_kernel(probe) || probe.conv.up_C != up.C) continue;   // the MFMA main kernel must take it
            cv.extra_in = {rz.in[0]};
            cv.name += " [+ upsample of " + rz.name + "]";
            plan.ops.erase(plan.ops.begin() + k);
            --k;
        }
    }

    // Which kernel runs a convolution, given the (resolved) strides / offsets and the dtypes of the tensors it touches: fills op.conv
    // and sets op.igemm when the implicit-GEMM MFMA kernel takes it (the direct kernel otherwise).  ONE predicate for finalize() and
    // for the int8 assignment, which may only put a tensor in int8 if every convolution touching it gets the MFMA path.
    bool choose_conv_kernel(POp& op) {
        ConvArgs& a = op.conv;
        const PTensor& ti = plan.tensors[op.in[0]];
        const PTensor& to = plan.tensors[op.out[0]];
        a.ld_in = ti.ld;
        a.ld_out = to.ld;
        a.ld_res = op.in.size() > 1 ? plan.tensors[op.in[1]].ld : 0;
        a.K = a.kh * a.kw * (a.Cin / a.groups);
        a.Cout_pad = a.Cout;
        a.Kpad = a.K;
        a.up_C = 0;
        if (!op.extra_in.empty()) {   // folded upsample: geometry of the half-resolution source
            const PTensor& tu = plan.tensors[op.extra_in[0]];
            a.up_C = tu.C;
            a.up_ld = tu.ld;
            a.up_H = tu.H;
            a.up_W = tu.W;
        }
        op.igemm = false;
        a.f32 = (op.stem && dt == DT_F32) ? 1 : 0;   // (the fp32 stem kernel; the MFMA path sets it below)
        if (op.kind == OP_CONV && !op.stem && dt == DT_F16 && a.groups == 1 && a.dil_h == 1 && a.dil_w == 1) {
            int cin_eff = a.Cin;
            bool ok = true;
            if (cin_eff % 8) {
                // padded channels are zero only for a freshly converted, un-aliased tensor
                const PTensor& own = plan.tensors[ti.parent >= 0 ? ti.parent : ti.id];
                ok = ti.parent < 0 && own.pad_zeroed;
                cin_eff = (a.Cin + 7) / 8 * 8;
            }
            ok = ok && ti.rcoff % 8 == 0 && ti.ld % 8 == 0;
            const bool in8 = ti.dtype == DT_I8, out8 = to.dtype == DT_I8;
            const bool res8 = op.in.size() > 1 && plan.tensors[op.in[1]].dtype == DT_I8;
            if (in8) ok = ok && a.Cin % 16 == 0 && ti.rcoff % 16 == 0 && ti.ld % 16 == 0;
            // output side: 16-byte stores when everything is a multiple of 8, element-wise stores otherwise
            bool vec_out = a.Cout % 8 == 0 && to.rcoff % 8 == 0 && to.ld % 8 == 0;
            if (op.in.size() > 1) {
                const PTensor& tr = plan.tensors[op.in[1]];
                vec_out = vec_out && tr.rcoff % 8 == 0 && tr.ld % 8 == 0;
            }
            // tiny reductions (K < 32, e.g. the DFL 1x1) stay on the direct kernel
            ok = ok && a.kh * a.kw * cin_eff >= 32;
            if (ok) {
                ConvArgs t = a;
                t.scalar_out = vec_out ? 0 : 1;
                t.Cin = cin_eff;
                t.bn = conv_igemm_pick_bn(t.Cout);
                t.bk = conv_igemm_pick_bk(cin_eff, t.kh * t.kw);
                {   // few tiles + long K at the largest batch: keep 32-wide steps so the wave-split-K variant applies
                    const long m_max = (long)(ti.nfix ? ti.nfix : plan.max_batch) * ti.nmul * t.Ho * t.Wo;
                    const long tiles128 = (m_max + 127) / 128 * ((t.Cout + t.bn - 1) / t.bn);
                    if (tiles128 <= 256 && (t.bn == 64 || t.bn == 80)) t.bk = 32;
                }
                t.CinK = conv_igemm_pick_cink(cin_eff, t.bk);  // a k-step never straddles a filter tap
                t.K = t.kh * t.kw * t.CinK;
                t.Kpad = (t.K + t.bk - 1) / t.bk * t.bk;
                t.bn = conv_igemm_pick_bn(t.Cout);
                t.Cout_pad = (t.Cout + t.bn - 1) / t.bn * t.bn;
                if (in8) {  // int8 operands: 64-channel k-steps; the input-side geometry is handed over in 2-byte units (ConvArgs)
                    t.bk = 32;
                    t.in_i8 = 1;
                    t.Cin = a.Cin / 2;
                    t.ld_in = ti.ld / 2;
                    t.CinK = (a.Cin + 63) / 64 * 64 / 2;
                    t.K = t.kh * t.kw * t.CinK;
                    t.Kpad = t.K;
                }
                t.out_i8 = out8 ? 1 : 0;
                t.res_i8 = res8 ? 1 : 0;
                t.out_inv_scale = out8 ? 1.0f / to.scale : 0.f;
                t.res_scale = res8 ? plan.tensors[op.in[1]].scale : 0.f;
                // Engines for several execution contexts in flight (setMaxAuxStreams(0)) take the implicit-GEMM operands through
                // registers instead of LDS-DMA: bit-identical results, and on YOLOv8n b32 with three contexts 34.2-34.5k img/s against
                // 33.3-33.6k (same box, alternating runs) - a DMA piece costs its wave 60-185 cycles of issue, which co-scheduled
                // workgroups of other contexts cannot hide for each other; a lone context is 3 % slower with it and keeps the DMA.
                // ... and only for layers below the MFMA / HBM ridge (312 FLOP per byte): the MFMA-bound GEMMs of res5 lose with it (C5 three
                // contexts 12.1 vs 11.65 ms; tools/gemm_tactics.py: K = 4608 0.39 vs 0.42 of peak), ResNet-50 gains 2.2 %, RetinaFace is
                // indifferent (profiles/r03_rs_ab.txt).
                {
                    const double flop_px = 2.0 * a.Cout * (double)a.kh * a.kw * a.Cin;
                    const double byte_px = 2.0 * ((double)a.Cin * a.stride_h * a.stride_w + (double)a.Cout * (op.in.size() > 1 ? 2 : 1));
                    constexpr double ridge = 312.0;   // the MFMA / HBM ridge, FLOP per byte
                    t.t_rs = (net.max_aux_streams == 0 && flop_px / byte_px < ridge) ? 1 : 0;
                }
                if (conv_igemm_supported(t)) {
                    a = t;
                    op.igemm = true;
                }
            }
        }
        // fp32 engines: the same skeleton on the fp32 MFMA (kernels/conv_igemm_f32.hip), 16-channel k-steps; the tile shape is the launcher's
        if (op.kind == OP_CONV && !op.stem && dt == DT_F32 && ti.dtype == DT_F32 && to.dtype == DT_F32 && a.groups == 1 && a.dil_h == 1 && a.dil_w == 1) {
            int cin_eff = a.Cin;
            bool ok = opt.f32_mfma;   // (TRTX_F32_DIRECT=1: the scalar direct kernel of rounds 1-4, an A/B switch)
            if (cin_eff % 4) {
                const PTensor& own = plan.tensors[ti.parent >= 0 ? ti.parent : ti.id];
                ok = ok && ti.parent < 0 && own.pad_zeroed;
                cin_eff = (a.Cin + 3) / 4 * 4;
            }
            ok = ok && ti.rcoff % 4 == 0 && ti.ld % 4 == 0;
            bool vec_out = a.Cout % 4 == 0 && to.rcoff % 4 == 0 && to.ld % 4 == 0;
            if (op.in.size() > 1) {
                const PTensor& tr = plan.tensors[op.in[1]];
                vec_out = vec_out && tr.rcoff % 4 == 0 && tr.ld % 4 == 0;
            }
            ok = ok && a.kh * a.kw * cin_eff >= 16 && a.Cout >= 8;   // tiny reductions / single-channel outputs (the DFL 1x1) stay on the direct kernel
            if (ok) {
                ConvArgs t = a;
                t.f32 = 1;
                t.scalar_out = vec_out ? 0 : 1;
                t.Cin = cin_eff;
                t.bk = 16;
                t.bn = 0;
                t.bm = 0;
                t.CinK = conv_igemm_f32_pick_cink(cin_eff);
                t.K = t.kh * t.kw * t.CinK;
                t.Kpad = (t.K + 15) / 16 * 16;
                t.Cout_pad = (t.Cout + 15) / 16 * 16;
                if (conv_igemm_f32_supported(t)) {
                    a = t;
                    op.igemm = true;
                }
            }
        }
        return op.igemm;
    }

    bool finalize() {
        // 0. view geometry (element units: independent of the dtypes assigned next)
        for (auto& t : plan.tensors) {
            int p = t.id, coff = 0;
            long eoff = 0;
            while (plan.tensors[p].parent >= 0) {
                coff += plan.tensors[p].coff;
                eoff += plan.tensors[p].eoff;
                p = plan.tensors[p].parent;
            }
            t.rcoff = coff;
            t.reoff = eoff;
            t.ld = plan.tensors[p].layout == LAY_NHWC ? plan.tensors[p].Calloc : 0;
        }
        fold_upsample();
        // kINT8: assign, then ask the kernel choice itself whether every convolution that touches an int8 tensor gets the MFMA path
        // (K >= 32, channel / offset / stride alignment, < 2 GB images ... - conditions the assignment's own screen does not repeat).
        // A convolution that does not takes its tensors out of the race and the assignment runs again; candidates only shrink, so
        // this ends, with such layers in fp16 as the builder flag promises (capi.cpp: "fall back to fp16").
        if (net.int8 && dt == DT_F16) {
            std::vector<char> veto(plan.tensors.size(), 0);
            auto top = [&](int t) {
                while (plan.tensors[t].parent >= 0) t = plan.tensors[t].parent;
                return t;
            };
            for (;;) {
                assign_int8(veto);
                bool again = false;
                for (const POp& op : plan.ops) {
                    if (op.kind != OP_CONV && op.kind != OP_DECONV) continue;
                    bool any8 = plan.tensors[op.out[0]].dtype == DT_I8;
                    for (int t : op.in) any8 = any8 || plan.tensors[t].dtype == DT_I8;
                    if (!any8) continue;
                    POp probe = op;
                    if (op.kind == OP_CONV && choose_conv_kernel(probe)) continue;
                    for (int t : op.in)
                        if (plan.tensors[t].dtype == DT_I8) veto[top(t)] = 1;
                    if (plan.tensors[op.out[0]].dtype == DT_I8) veto[top(op.out[0])] = 1;
                    again = true;
                }
                if (!again) break;
                for (PTensor& t : plan.tensors)
                    if (t.dtype == DT_I8) {
                        t.dtype = DT_F16;
                        t.scale = 0.f;
                    }
            }
        }
        // 1. storages for owners
        for (auto& t : plan.tensors) {
            if (t.parent >= 0 || t.storage >= 0) continue;
            Storage s;
            s.kind = ST_ARENA;
            s.bytes = tensor_bytes(t);
            t.storage = (int)plan.storages.size();
            plan.storages.push_back(s);
        }
        // 2. views share their owner's storage
        for (auto& t : plan.tensors) {
            int p = t.id;
            while (plan.tensors[p].parent >= 0) p = plan.tensors[p].parent;
            t.storage = plan.tensors[p].storage;
        }
        // binding storages take their size from the bound tensor
        for (size_t b = 0; b < plan.binding_ptensor.size(); ++b) {
            const PTensor& t = plan.tensors[plan.binding_ptensor[b]];
            plan.storages[t.storage].bytes = tensor_bytes(t);
        }
        // 3. convolution kernel selection now that strides/offsets are known
        for (auto& op : plan.ops) {
            if (op.kind != OP_CONV && op.kind != OP_DECONV) continue;
            ConvArgs& a = op.conv;
            const PTensor& ti = plan.tensors[op.in[0]];
            const PTensor& to = plan.tensors[op.out[0]];
            if (op.stem && (dt == DT_F16 ? (to.ld % 8 || to.rcoff % 8) : (to.ld % 4 || to.rcoff % 4))) return fail(op.name + ": stem convolution output is not 16-byte aligned");
            choose_conv_kernel(op);
            if (!op.extra_in.empty() && !op.igemm) return fail(op.name + ": folded upsample on a convolution that cannot take the MFMA path");
            if ((ti.dtype == DT_I8 || to.dtype == DT_I8 || (op.in.size() > 1 && plan.tensors[op.in[1]].dtype == DT_I8)) && !op.igemm)
                return fail(op.name + ": int8 tensor on a convolution that cannot take the MFMA path");
            const double es_in = (double)dtype_size(ti.dtype), es_out = (double)dtype_size(to.dtype);
            const double cin_real = ti.dtype == DT_I8 ? 2.0 * a.Cin : (double)a.Cin;
            // (a folded upsample reads up_C of its input channels from a tensor a quarter the size)
            const double in_elems = (double)a.H * a.W * (cin_real - a.up_C) + (double)a.up_H * a.up_W * a.up_C;
            op.bytes = to.nmul * ((op.stem ? 4.0 : es_in) * in_elems + es_out * (double)a.Ho * a.Wo * a.Cout * (op.in.size() > 1 ? 2 : 1));
        }
        // 4. plugins: configure + workspace
        for (size_t k = 0; k < plan.ops.size(); ++k) {
            POp& op = plan.ops[k];
            if (op.kind != OP_PLUGIN) continue;
            std::vector<trtx_dims> din, dout;
            for (int t : op.in) din.push_back(to_c(plan.tensors[t].dims));
            for (int t : op.out) dout.push_back(to_c(plan.tensors[t].dims));
            if (op.plugin->v.configure &&
                op.plugin->v.configure(op.plugin->v.self, din.data(), (int)din.size(), dout.data(), (int)dout.size(),
                                       plan.max_batch) != 0)
                return fail(op.name + ": plugin configurePlugin rejected the tensor shapes");
            op.ws_bytes = op.plugin->v.workspace_size ? op.plugin->v.workspace_size(op.plugin->v.self, plan.max_batch) : 0;
            for (int t : op.in) op.bytes += 4.0 * plan.tensors[t].dims.volume();
            for (int t : op.out) op.bytes += 4.0 * plan.tensors[t].dims.volume();
        }
        // SPPF (yolov8/src/block.cpp:214-237): y1 = maxpool_k(x), y2 = maxpool_k(y1), y3 = maxpool_k(y2), stride 1,
        // 'same' padding.  Three tiny launches (20x20 maps) become one that keeps the map in LDS.
        for (size_t k = 0; k + 2 < plan.ops.size(); ++k) {
            auto same_max = [&](const POp& o) {
                return o.kind == OP_POOL && o.i[0] == POOL_MAX && o.i[1] == o.i[2] && (o.i[1] & 1) && o.i[3] == 1 && o.i[4] == 1 &&
                       o.i[5] == o.i[1] / 2 && o.i[6] == o.i[1] / 2;
            };
            POp &a = plan.ops[k], &b = plan.ops[k + 1], &c3 = plan.ops[k + 2];
            if (!same_max(a) || !same_max(b) || !same_max(c3) || a.i[1] != b.i[1] || a.i[1] != c3.i[1]) continue;
            if (b.in[0] != a.out[0] || c3.in[0] != b.out[0] || (dt != DT_F16 && dt != DT_F32)) continue;
            const PTensor& tx = plan.tensors[a.in[0]];
            const int cv = dt == DT_F16 ? 8 : 4;   // channels per 16-byte chunk (round 6: fp32 engines too)
            bool ok = tx.C % cv == 0 && tx.ld % cv == 0 && tx.rcoff % cv == 0 && (long)(tx.H + a.i[1] - 1) * (tx.W + a.i[1] - 1) * 32 <= 64 * 1024 && tx.nmul == 1;
            for (const POp* o : {&a, &b, &c3}) {
                const PTensor& ty = plan.tensors[o->out[0]];
                ok = ok && ty.ld % cv == 0 && ty.rcoff % cv == 0 && ty.H == tx.H && ty.W == tx.W && ty.C == tx.C;
            }
            if (!ok) continue;
            a.kind = OP_POOL_CHAIN;
            a.name += " [x3 chained]";
            a.out = {a.out[0], b.out[0], c3.out[0]};
            plan.ops.erase(plan.ops.begin() + k + 1, plan.ops.begin() + k + 3);
        }
        group_convs();
        // 5. op dependencies at (storage, channel/element range) granularity: RAW, WAR and WAW
        const int nops = (int)plan.ops.size();
        struct Access { int storage; long lo, hi; int op; bool write; };
        auto access_of = [&](int t, int op, bool write) {
            const PTensor& pt = plan.tensors[t];
            Access a{pt.storage, 0, 0, op, write};
            if (pt.layout == LAY_NHWC) {
                a.lo = pt.rcoff;
                a.hi = pt.rcoff + (pt.parent < 0 && pt.Calloc > pt.C ? pt.Calloc : pt.C);  // the layout pass zero-fills its padding
            } else {
                a.lo = pt.reoff;
                a.hi = pt.reoff + pt.dims.volume();
            }
            return a;
        };
        std::vector<std::vector<int>> deps(nops);
        {
            std::vector<Access> log;
            for (int k = 0; k < nops; ++k) {
                const POp& op = plan.ops[k];
                std::vector<Access> mine;
                for (int t : op.in) mine.push_back(access_of(t, k, false));
                for (int t : op.extra_in) mine.push_back(access_of(t, k, false));
                for (int t : op.out) mine.push_back(access_of(t, k, true));
                for (const Access& m : mine)
                    for (const Access& o : log)
                        if (o.storage == m.storage && o.lo < m.hi && m.lo < o.hi && (o.write || m.write) && o.op != k)
                            deps[k].push_back(o.op);
                std::sort(deps[k].begin(), deps[k].end());
                deps[k].erase(std::unique(deps[k].begin(), deps[k].end()), deps[k].end());
                log.insert(log.end(), mine.begin(), mine.end());
            }
        }
        // 6. lanes (= HIP streams at run time): an op continues the lane of a dependency that is still that lane's tail,
        // otherwise it opens a free lane, otherwise it queues behind the lane that went idle first.  Independent branches
        // (the six cv2/cv3 head chains of YOLOv8, model.cpp:224-291; FPN/SSH branches of RetinaFace) end up on different
        // lanes and overlap on the GPU; a sequential network stays on lane 0.
        // At most 4 by default: HIP multiplexes all streams of a process onto 4 hardware queues (GPU_MAX_HW_QUEUES), so further lanes
        // only alias with each other and with the caller's own copy / post-processing streams.  Measured (YOLOv8n b32, same box):
        // 6 lanes 1.335-1.353 ms vs 4 lanes 1.354-1.360 with resident inputs, but 2.56-2.74 vs 1.67-1.71 ms once a host-fed pipeline
        // adds an H2D stream; ResNet-50 / RetinaFace / R-CNN are 0.5-1.3 % faster with 4.  (8 or 16 hardware queues: 2.2x slower.)
        int max_lanes = net.max_aux_streams >= 0 ? 1 + net.max_aux_streams : 4;  // IBuilderConfig::setMaxAuxStreams
        if (opt.lanes > 0) max_lanes = std::max(1, std::min(16, opt.lanes));  // A/B override (TRTX_LANES)
        std::vector<int> tail(max_lanes, -1);
        plan.num_lanes = 1;
        for (int k = 0; k < nops; ++k) {
            int lane = -1;
            for (int d : deps[k])
                if (tail[plan.ops[d].lane] == d && (lane < 0 || d > tail[lane])) lane = plan.ops[d].lane;
            if (lane < 0 && deps[k].empty()) lane = 0;  // sources (input conversions) stay on the caller's stream
            if (lane < 0) {
                for (int l = 0; l < max_lanes && lane < 0; ++l)
                    if (tail[l] < 0) lane = l;
            }
            if (lane < 0) {
                lane = 0;
                for (int l = 1; l < max_lanes; ++l)
                    if (tail[l] < tail[lane]) lane = l;
            }
            plan.ops[k].lane = lane;
            tail[lane] = k;
            plan.num_lanes = std::max(plan.num_lanes, lane + 1);
        }
        // happens-before closure over dependency edges and lane order
        const int words = (nops + 63) / 64;
        std::vector<std::vector<uint64_t>> anc(nops, std::vector<uint64_t>(words, 0));
        {
            std::vector<int> prev_on_lane(max_lanes, -1);
            for (int k = 0; k < nops; ++k) {
                auto absorb = [&](int d) {
                    for (int w = 0; w < words; ++w) anc[k][w] |= anc[d][w];
                    anc[k][d >> 6] |= 1ull << (d & 63);
                };
                for (int d : deps[k]) absorb(d);
                const int lane = plan.ops[k].lane;
                if (prev_on_lane[lane] >= 0) absorb(prev_on_lane[lane]);
                prev_on_lane[lane] = k;
            }
            // cross-lane waits: a dependency on another lane needs an event unless an earlier wait already covers it
            std::vector<std::vector<int>> covered(max_lanes, std::vector<int>(max_lanes, -1));
            for (int k = 0; k < nops; ++k) {
                POp& op = plan.ops[k];
                for (int d : deps[k]) {
                    const int ld = plan.ops[d].lane;
                    if (ld == op.lane || covered[op.lane][ld] >= d) continue;
                    op.wait_ops.push_back(d);
                    plan.ops[d].signal = true;
                    covered[op.lane][ld] = d;
                }
            }
        }
        auto before = [&](int a, int b) { return a == b || ((anc[b][a >> 6] >> (a & 63)) & 1ull); };
        // 7. which ops touch which arena storage
        std::vector<std::vector<int>> touch(plan.storages.size());
        for (int k = 0; k < nops; ++k) {
            const POp& op = plan.ops[k];
            auto mark = [&](int t) {
                const int st = plan.tensors[t].storage;
                Storage& s = plan.storages[st];
                s.first_use = std::min(s.first_use, k);
                s.last_use = std::max(s.last_use, k);
                if (touch[st].empty() || touch[st].back() != k) touch[st].push_back(k);
            };
            for (int t : op.in) mark(t);
            for (int t : op.extra_in) mark(t);
            for (int t : op.out) mark(t);
        }
        // plugin workspaces are short-lived arena blocks
        std::vector<std::pair<int, int>> ws_storage;  // (op, storage)
        for (int k = 0; k < nops; ++k) {
            if ((plan.ops[k].kind != OP_PLUGIN && plan.ops[k].kind != OP_YOLO_HEAD) || plan.ops[k].ws_bytes == 0) continue;
            Storage s;
            s.kind = ST_ARENA;
            s.bytes = plan.ops[k].ws_bytes;
            s.first_use = s.last_use = k;
            ws_storage.push_back({k, (int)plan.storages.size()});
            plan.storages.push_back(s);
            touch.push_back({k});
        }
        // 8. arena offsets, first fit.  Two blocks may share memory only if every op touching one happens-before every op
        // touching the other (with one lane this is the classic disjoint-live-interval rule).
        auto ordered = [&](int sa, int sb) {
            for (int x : touch[sa])
                for (int y : touch[sb])
                    if (!before(x, y) || x == y) return false;
            return true;
        };
        std::vector<int> order;
        for (size_t si = 0; si < plan.storages.size(); ++si)
            if (plan.storages[si].kind == ST_ARENA && plan.storages[si].last_use >= 0) order.push_back((int)si);
        std::stable_sort(order.begin(), order.end(),
                         [&](int x, int y) { return plan.storages[x].first_use < plan.storages[y].first_use; });
        std::vector<int> placed;
        size_t arena = 0;
        for (int si : order) {
            Storage& st = plan.storages[si];
            const size_t need = align_up256(st.bytes);
            std::vector<std::pair<size_t, size_t>> busy;
            for (int pj : placed) {
                const Storage& o = plan.storages[pj];
                if (ordered(pj, si) || ordered(si, pj)) continue;
                busy.push_back({o.offset, o.offset + align_up256(o.bytes)});
            }
            std::sort(busy.begin(), busy.end());
            size_t off = 0;
            for (auto& bz : busy) {
                if (off + need <= bz.first) break;
                off = std::max(off, bz.second);
            }
            st.offset = off;
            arena = std::max(arena, off + need);
            placed.push_back(si);
        }
        plan.arena_bytes = arena;
        for (auto& w : ws_storage) plan.ops[w.first].ws_off = plan.storages[w.second].offset;
        return true;
    }
    static size_t align_up256(size_t v) { return (v + 255) / 256 * 256; }
};

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

static thread_local int g_calibration_lowering = 0;
CalibrationLowering::CalibrationLowering() { ++g_calibration_lowering; }
CalibrationLowering::~CalibrationLowering() { --g_calibration_lowering; }
bool CalibrationLowering::active() { return g_calibration_lowering > 0; }

bool lower_network(const Network& net, Plan* plan) {
    *plan = Plan();
    Lowerer L(net, *plan);
    const bool ok = L.run();
    if (!ok) plan->error = L.err.empty() ? "lowering failed" : L.err;
    return ok;
}

// ---------------------------------------------------------------------------------------------------------
bool pack_weights(const Network& net, Plan* plan) {
    std::vector<uint8_t>& blob = plan->weight_blob;
    blob.clear();
    auto reserve = [&](size_t bytes) {
        const size_t off = align256(blob.size());
        blob.resize(off + bytes, 0);
        return off;
    };
    // constants
    for (auto& t : plan->tensors) {
        if (t.storage < 0 || plan->storages[t.storage].kind != ST_WEIGHTS || t.parent >= 0) continue;
        const TensorDef& nt = net.tensors[t.net_tensor];
        const LayerDef& l = net.layers[nt.producer];
        const size_t off = reserve(l.w0.size() * 4);
        memcpy(blob.data() + off, l.w0.data(), l.w0.size() * 4);
        plan->storages[t.storage].offset = off;
    }
    std::vector<POp*> every;   // the members of a conv group are packed like the convolutions they are
    for (auto& op : plan->ops) {
        for (auto& m : op.group) every.push_back(&m);
        every.push_back(&op);
    }
    for (POp* pop : every) {
        POp& op = *pop;
        if (op.kind == OP_CONV || op.kind == OP_DECONV) {
            const LayerDef& l = net.layers[op.src_layer];
            const ConvArgs& a = op.conv;
            const int cout = a.Cout;
            // folded per-channel scale / shift
            std::vector<float> sc(cout, 1.f), bias(std::max(a.Cout_pad, cout), 0.f);
            for (int c = 0; c < cout && c < (int)l.w1.size(); ++c) bias[c] = l.w1[c];
            if (op.scale_layer >= 0) {
                const LayerDef& s = net.layers[op.scale_layer];
                for (int c = 0; c < cout; ++c) {
                    const float scale = s.w1.empty() ? 1.f : (s.w1.size() == 1 ? s.w1[0] : s.w1[c]);
                    const float shift = s.w0.empty() ? 0.f : (s.w0.size() == 1 ? s.w0[0] : s.w0[c]);
                    sc[c] = scale;
                    bias[c] = bias[c] * scale + shift;
                }
            }
            const TensorDef& tin = net.tensors[l.inputs[0]];
            const int cin_logical = (int)tin.dims.d[tin.dims.nb - 3];
            if (op.kind == OP_DECONV) {
                op.w_off = reserve((size_t)cout * a.kh * a.kw * (cin_logical / a.groups) * 4);
                pack_deconv_weights_f32(l.w0.data(), cin_logical, cout, a.groups, a.kh, a.kw,
                                        reinterpret_cast<float*>(blob.data() + op.w_off));
            } else if (op.stem) {
                // [tap = (c*kh + r)*kw + q][cout], BN scale folded
                op.w_off = reserve((size_t)a.kh * a.kw * cin_logical * cout * 4);
                float* dst = reinterpret_cast<float*>(blob.data() + op.w_off);
                for (int co = 0; co < cout; ++co)
                    for (int t = 0; t < cin_logical * a.kh * a.kw; ++t)
                        dst[(size_t)t * cout + co] = l.w0[(size_t)co * cin_logical * a.kh * a.kw + t] * sc[co];
            } else if (op.from_deconv) {
                // CKRS [Cin][Cout][kh][kw] -> KCRS of the stand-in 1x1 conv: output channel (r*kw + q)*Cout + co.  The re-layout
                // (and the per-sub-position bias) does not depend on which conv kernel runs the stand-in: finalize() may refuse
                // the MFMA path (kh*kw*Cin < 32, > 2 GB images) and the direct kernel must then see the same KCRS weights.
                const int taps = l.kernel[0] * l.kernel[1], dc = l.nb_out;
                std::vector<float> w2((size_t)cout * cin_logical);
                for (int ci = 0; ci < cin_logical; ++ci)
                    for (int co = 0; co < dc; ++co)
                        for (int t = 0; t < taps; ++t) w2[(size_t)(t * dc + co) * cin_logical + ci] = l.w0[((size_t)ci * dc + co) * taps + t];
                for (int c = 0; c < cout; ++c) bias[c] = l.w1.empty() ? 0.f : l.w1[c % dc];
                if (op.igemm) {
                    op.w_off = reserve((size_t)a.Cout_pad * a.Kpad * 2);
                    pack_conv_weights_f16(w2.data(), cout, cin_logical, 1, 1, a.CinK, a.bk, sc.data(),
                                          reinterpret_cast<uint16_t*>(blob.data() + op.w_off));
                } else {
                    op.w_off = reserve((size_t)cout * cin_logical * 4);
                    pack_conv_weights_f32(w2.data(), cout, cin_logical, 1, 1, sc.data(), reinterpret_cast<float*>(blob.data() + op.w_off));
                }
            } else if (op.igemm && a.in_i8) {
                // int8 weights, per-output-channel scales; cscale[c] = input tensor scale * weight scale (dequantises the int32 sums)
                const size_t kpad_bytes = (size_t)a.Kpad * 2;
                op.w_off = reserve((size_t)a.Cout_pad * kpad_bytes);
                std::vector<float> wscale(a.Cout_pad, 1.f);
                conv_pack_weights_i8(l.w0.data(), cout, cin_logical, a.kh, a.kw, a.CinK * 2, sc.data(), a.Cout_pad, (int)kpad_bytes,
                                     reinterpret_cast<int8_t*>(blob.data() + op.w_off), wscale.data());
                const float s_in = plan->tensors[op.in[0]].scale;
                for (float& v : wscale) v *= s_in;
                op.s_off = reserve(wscale.size() * 4);
                memcpy(blob.data() + op.s_off, wscale.data(), wscale.size() * 4);
            } else if (op.igemm && a.f32) {
                op.w_off = reserve((size_t)a.Cout_pad * a.Kpad * 4);
                conv_pack_weights_igemm_f32(l.w0.data(), cout, cin_logical, a.kh, a.kw, a.CinK, a.Kpad, a.Cout_pad, sc.data(),
                                            reinterpret_cast<float*>(blob.data() + op.w_off));
            } else if (op.igemm) {
                op.w_off = reserve((size_t)a.Cout_pad * a.Kpad * 2);
                pack_conv_weights_f16(l.w0.data(), cout, cin_logical, a.kh, a.kw, a.CinK, a.bk, sc.data(),
                                      reinterpret_cast<uint16_t*>(blob.data() + op.w_off));
            } else {
                op.w_off = reserve((size_t)cout * a.kh * a.kw * (cin_logical / a.groups) * 4);
                pack_conv_weights_f32(l.w0.data(), cout, cin_logical / a.groups, a.kh, a.kw, sc.data(),
                                      reinterpret_cast<float*>(blob.data() + op.w_off));
            }
            op.b_off = reserve(bias.size() * 4);
            memcpy(blob.data() + op.b_off, bias.data(), bias.size() * 4);
            op.bytes += (double)(op.igemm ? (size_t)a.Cout_pad * a.Kpad * (a.f32 ? 4 : 2) : (size_t)cout * a.K * 4);
        } else if (op.kind == OP_YOLO_HEAD) {
            const LayerDef& l = net.layers[op.src_layer];
            op.w_off = reserve(16 * 4);
            memcpy(blob.data() + op.w_off, l.w0.data(), 16 * 4);
        } else if (op.kind == OP_SCALE_NHWC || op.kind == OP_SCALE_LIN) {
            const LayerDef& l = net.layers[op.src_layer];
            const int C = op.kind == OP_SCALE_NHWC ? plan->tensors[op.in[0]].C : (op.i[0] == 1 ? op.i[2] : 1);
            auto expand = [&](const std::vector<float>& w, float dflt) {
                std::vector<float> v(C, dflt);
                for (int c = 0; c < C; ++c)
                    if (!w.empty()) v[c] = w.size() == 1 ? w[0] : w[c];
                return v;
            };
            const auto shift = expand(l.w0, 0.f), scale = expand(l.w1, 1.f), power = expand(l.w2, 1.f);
            op.s_off = reserve(C * 4);
            memcpy(blob.data() + op.s_off, scale.data(), C * 4);
            op.b_off = reserve(C * 4);
            memcpy(blob.data() + op.b_off, shift.data(), C * 4);
            op.w_off = reserve(C * 4);
            memcpy(blob.data() + op.w_off, power.data(), C * 4);
        }
    }
    for (auto& op : plan->ops)   // a grouped launch moves what its members move (their packed weights were priced just above)
        if (op.kind == OP_CONV_GROUP) {
            op.bytes = 0;
            for (const POp& m : op.group) op.bytes += m.bytes;
        }
    plan->weight_bytes = align256(blob.size());
    blob.resize(plan->weight_bytes, 0);
    return true;
}

// ---------------------------------------------------------------------------------------------------------
std::string Plan::describe_json() const {
    std::ostringstream o;
    double flops = 0, bytes = 0;
    int n_conv = 0, n_igemm = 0;
    for (const auto& op : ops) {
        flops += op.flops;
        bytes += op.bytes;
        if (op.kind == OP_CONV) {
            ++n_conv;
            n_igemm += op.igemm ? 1 : 0;
        }
        n_conv += (int)op.group.size();
        n_igemm += (int)op.group.size();
    }
    o << "{\"fp16\":" << (fp16 ? "true" : "false") << ",\"max_batch\":" << max_batch << ",\"arena_bytes\":" << arena_bytes
      << ",\"weight_bytes\":" << weight_bytes << ",\"n_lanes\":" << num_lanes << ",\"n_ops\":" << ops.size() << ",\"n_conv\":" << n_conv
      << ",\"n_igemm\":" << n_igemm << ",\"flops_per_sample\":" << flops << ",\"bytes_per_sample\":" << bytes
      << ",\"ops\":[";
    for (size_t k = 0; k < ops.size(); ++k) {
        const POp& op = ops[k];
        o << (k ? "," : "") << "{\"kind\":\"" << op_kind_name(op.kind) << "\",\"name\":\"";
        for (char c : op.name) o << ((c == '"' || c == '\\' || (unsigned char)c < 0x20) ? ' ' : c);
        o << "\",\"flops\":" << op.flops << ",\"bytes\":" << op.bytes;
        auto conv_fields = [&](const POp& op) {
            const ConvArgs& a = op.conv;
            o << ",\"igemm\":" << (op.igemm ? "true" : "false") << ",\"stem\":" << (op.stem ? "true" : "false") << ",\"cin\":" << (a.in_i8 ? 2 * a.Cin : a.Cin) << ",\"cout\":" << a.Cout
              << ",\"k\":[" << a.kh << "," << a.kw << "],\"stride\":[" << a.stride_h << "," << a.stride_w << "],\"hw_in\":["
              << a.H << "," << a.W << "],\"hw_out\":[" << a.Ho << "," << a.Wo << "],\"act1\":" << a.act1
              << ",\"act2\":" << a.act2 << ",\"alpha1\":" << a.alpha1 << ",\"alpha2\":" << a.alpha2 << ",\"up_c\":" << a.up_C << ",\"residual\":" << (op.in.size() > 1 ? "true" : "false")
              << ",\"bn_folded\":" << (op.scale_layer >= 0 ? "true" : "false") << ",\"ld_in\":" << a.ld_in
              << ",\"ld_out\":" << a.ld_out << ",\"i8\":[" << a.in_i8 << "," << a.out_i8 << "," << a.res_i8 << "],\"nmul\":" << (op.stem ? 1 : tensors[op.in[0]].nmul) << ",\"nfix\":"
              << (op.stem ? 0 : tensors[op.in[0]].nfix);
        };
        if (op.kind == OP_CONV || op.kind == OP_DECONV) conv_fields(op);
        if (op.kind == OP_CONV_GROUP) {
            o << ",\"members\":[";
            for (size_t j = 0; j < op.group.size(); ++j) {
                const POp& m = op.group[j];
                o << (j ? "," : "") << "{\"name\":\"";
                for (char c : m.name) o << ((c == '"' || c == '\\' || (unsigned char)c < 0x20) ? ' ' : c);
                o << "\",\"flops\":" << m.flops << ",\"bytes\":" << m.bytes;
                conv_fields(m);
                o << ",\"in\":[";
                for (size_t q = 0; q < m.in.size(); ++q) o << (q ? "," : "") << m.in[q];
                o << "],\"out\":[" << m.out[0] << "]}";
            }
            o << "]";
        }
        o << ",\"lane\":" << op.lane << ",\"waits\":[";
        for (size_t j = 0; j < op.wait_ops.size(); ++j) o << (j ? "," : "") << op.wait_ops[j];
        o << "],\"in\":[";
        for (size_t j = 0; j < op.in.size(); ++j) o << (j ? "," : "") << op.in[j];
        o << "],\"out\":[";
        for (size_t j = 0; j < op.out.size(); ++j) o << (j ? "," : "") << op.out[j];
        o << "]}";
    }
    o << "],\"tensors\":[";
    for (size_t k = 0; k < tensors.size(); ++k) {
        const PTensor& t = tensors[k];
        o << (k ? "," : "") << "{\"id\":" << t.id << ",\"net\":" << t.net_tensor << ",\"layout\":\""
          << (t.layout == LAY_NHWC ? "nhwc" : "linear") << "\",\"dims\":[";
        for (int d = 0; d < t.dims.nb; ++d) o << (d ? "," : "") << t.dims.d[d];
        o << "],\"storage\":" << t.storage << ",\"coff\":" << t.rcoff << ",\"ld\":" << t.ld << ",\"dtype\":" << t.dtype << ",\"scale\":" << t.scale
          << ",\"view\":"
          << (t.parent >= 0 ? "true" : "false") << "}";
    }
    o << "],\"storages\":[";
    for (size_t k = 0; k < storages.size(); ++k) {
        const Storage& s = storages[k];
        o << (k ? "," : "") << "{\"kind\":" << s.kind << ",\"bytes\":" << s.bytes << ",\"offset\":" << s.offset
          << ",\"first\":" << s.first_use << ",\"last\":" << s.last_use << "}";
    }
    o << "]}";
    return o.str();
}

}  // namespace trtx
