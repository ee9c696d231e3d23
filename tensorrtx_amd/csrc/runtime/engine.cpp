#include "engine.h"

#include <stdlib.h>
#include <string.h>

#include <sstream>

#include "../common.h"
#include "../options.h"
#include "int8.h"
#include "plugin.h"

using namespace trtx;

trtx_engine::~trtx_engine() {
    int left = plugins_initialized;  // initialize() ran on the first `plugins_initialized` plugin ops, in plan order
    for (auto& op : plan.ops)
        if (op.kind == OP_PLUGIN && left-- > 0 && op.plugin->v.terminate) op.plugin->v.terminate(op.plugin->v.self);
    if (d_weights) (void)hipFree(d_weights);
}

trtx_context::~trtx_context() {
    for (hipStream_t st : lane_stream)
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (hipEvent_t ev : op_event)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : lane_done)
        if (ev) (void)hipEventDestroy(ev);
    if (start_event) (void)hipEventDestroy(start_event);
    for (auto& g : graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    if (d_arena) (void)hipFree(d_arena);
}

namespace trtx {

namespace {

struct Resolver {
    const Plan& plan;
    char* arena;
    char* weights;
    void* const* bindings;
    char* base(const PTensor& t) const {
        const Storage& s = plan.storages[t.storage];
        switch (s.kind) {
            case ST_ARENA: return arena + s.offset;
            case ST_WEIGHTS: return weights + s.offset;
            default: return static_cast<char*>(bindings[s.binding]);
        }
    }
    void* ptr(int tid) const {
        const PTensor& t = plan.tensors[tid];
        const size_t es = dtype_size(t.dtype);
        return base(t) + (t.layout == LAY_NHWC ? (size_t)t.rcoff : (size_t)t.reoff) * es;
    }
};

// add the runtime batch as outermost dimension of a per-sample strided view
StridedView batch_view(const StridedView& v, int batch, long bs_in, long bs_in2) {
    StridedView o{};
    o.rank = v.rank + 1;
    o.shape[0] = batch;
    o.stride_in[0] = bs_in;
    o.stride_in2[0] = bs_in2;
    for (int d = 0; d < v.rank; ++d) {
        o.shape[d + 1] = v.shape[d];
        o.stride_in[d + 1] = v.stride_in[d];
        o.stride_in2[d + 1] = v.stride_in2[d];
    }
    return o;
}

}  // namespace

int32_t execute_plan(trtx_context* c, int batch, void* const* bindings, hipStream_t stream,
                     std::vector<OpTiming>* prof) {
    trtx_engine* e = c->engine;
    const Plan& plan = e->plan;
    if (batch < 1 || batch > plan.max_batch) {
        fprintf(stderr, "[trtx_hip] enqueue: batch %d outside [1, %d]\n", batch, plan.max_batch);
        return TRTX_ERR_INVALID;
    }
    for (size_t b = 0; b < plan.binding_tensor.size(); ++b)
        if (!bindings[b] && !(c->frames && plan.binding_is_input[b])) {   // enqueue_frames: the input tensor is never touched
            fprintf(stderr, "[trtx_hip] enqueue: binding %zu is null\n", b);
            return TRTX_ERR_INVALID;
        }
    Resolver R{plan, static_cast<char*>(c->d_arena), static_cast<char*>(e->d_weights), bindings};
    char* W = static_cast<char*>(e->d_weights);
    std::vector<hipEvent_t> evs;
    std::vector<LaunchProbe> probes;  // profiling: per convolution, the dispatch's own begin / end timestamps
    const bool no_probe = !options().profile_kernel_events;
    if (prof) {
        evs.resize(plan.ops.size() + 1);
        for (auto& ev : evs) TRTX_HIP_TRY(hipEventCreate(&ev));
        probes.resize(plan.ops.size());
        if (!no_probe && !c->tuning)  // the tactic timing keeps its own clock (the interval between the stream events)
            for (size_t k = 0; k < plan.ops.size(); ++k)
                if ((plan.ops[k].kind == OP_CONV && plan.ops[k].igemm) || plan.ops[k].kind == OP_CONV_GROUP)
                    if (hipEventCreate(&probes[k].start) != hipSuccess || hipEventCreate(&probes[k].stop) != hipSuccess) (void)hipGetLastError();
        TRTX_HIP_TRY(hipEventRecord(evs[0], stream));
    }
    auto free_probes = [&]() {
        conv_set_launch_probe(nullptr);
        for (auto& pr : probes) {
            if (pr.start) (void)hipEventDestroy(pr.start);
            if (pr.stop) (void)hipEventDestroy(pr.stop);
        }
        probes.clear();
    };
    // lanes: independent branches of the plan run on the context's own streams, fenced by events (profiling runs
    // everything on the caller's stream so that the per-op events measure isolated kernels)
    hipStream_t const user_stream = stream;
    const bool lanes = !prof && !c->observer && plan.num_lanes > 1;  // calibration statistics are collected on one stream
    std::vector<char> lane_started(plan.num_lanes, 0);
    // every failure from here on leaves through this: nothing keeps running behind the caller's back (each lane that was started joins
    // the caller's stream) and the profiling events are released
    auto bail = [&](int32_t st) {
        (void)hipGetLastError();
        if (lanes)
            for (int l = 1; l < plan.num_lanes; ++l)
                if (lane_started[l] && hipEventRecord(c->lane_done[l], c->lane_stream[l]) == hipSuccess)
                    (void)hipStreamWaitEvent(user_stream, c->lane_done[l], 0);
        for (auto& ev : evs) (void)hipEventDestroy(ev);
        evs.clear();
        free_probes();
        return st;
    };
    if (lanes && hipEventRecord(c->start_event, user_stream) != hipSuccess) return bail(TRTX_ERR_HIP);
    for (size_t k = 0; k < plan.ops.size(); ++k) {
        const POp& op = plan.ops[k];
        hipStream_t stream = user_stream;  // shadows the parameter: the stream THIS op is issued on
        if (lanes) {
            if (op.lane > 0) {
                stream = c->lane_stream[op.lane];
                if (!lane_started[op.lane]) {
                    if (hipStreamWaitEvent(stream, c->start_event, 0) != hipSuccess) return bail(TRTX_ERR_HIP);
                    lane_started[op.lane] = 1;
                }
            }
            for (int d : op.wait_ops)
                if (hipStreamWaitEvent(stream, c->op_event[d], 0) != hipSuccess) return bail(TRTX_ERR_HIP);
        }
        const PTensor& t0 = plan.tensors[op.in.empty() ? op.out[0] : op.in[0]];
        const PTensor& to = plan.tensors[op.out[0]];
        auto nb = [&](const PTensor& t) { return (t.nfix ? t.nfix : batch) * t.nmul; };
        int32_t st = TRTX_OK;
        const bool skip = c->tuning && (op.kind == OP_PLUGIN || op.kind == OP_YOLO_HEAD || op.kind == OP_ROI_ALIGN);
        if (!skip) switch (op.kind) {
            case OP_CONV:
            case OP_DECONV: {
                ConvArgs a = op.conv;
                a.in = R.ptr(op.in[0]);
                a.out = R.ptr(op.out[0]);
                a.residual = op.in.size() > 1 ? R.ptr(op.in[1]) : nullptr;
                a.up_in = op.extra_in.empty() ? nullptr : R.ptr(op.extra_in[0]);   // folded upsample: the half-resolution source
                a.wgt = W + op.w_off;
                a.bias = reinterpret_cast<const float*>(W + op.b_off);
                a.cscale = a.in_i8 ? reinterpret_cast<const float*>(W + op.s_off) : nullptr;
                a.N = op.stem ? batch : nb(t0);
                a.M = a.N * a.Ho * a.Wo;
                if (op.kind == OP_DECONV)
                    st = deconv_direct(a, op.dtype, stream);
                else if (op.stem && c->frames)
                    st = a.f32 ? TRTX_ERR_UNSUPPORTED : conv_stem_frames_f32(a, c->frames, stream);   // letterbox fused into the stem (trtx_context_enqueue_frames; kFP16 engines)
                else if (op.stem)
                    st = a.f32 ? conv_stem_nchw_f32_out_f32(a, stream) : conv_stem_nchw_f32(a, stream);
                else if (op.igemm) {
                    if (prof && probes[k].start && probes[k].stop) conv_set_launch_probe(&probes[k]);
                    st = a.f32 ? conv_igemm_f32(a, stream) : conv_igemm_f16(a, stream);
                    conv_set_launch_probe(nullptr);
                } else
                    st = conv_direct(a, op.dtype, stream);
                break;
            }
            case OP_CONV_GROUP: {   // independent convolutions of one kernel instantiation: one launch (conv_igemm_group_f16_kernel)
                ConvArgs ga[kMaxConvGroup];
                const int gn = (int)op.group.size();
                if (gn < 2 || gn > kMaxConvGroup) { st = TRTX_ERR_STATE; break; }
                for (int m = 0; m < gn; ++m) {
                    const POp& mo = op.group[m];
                    ConvArgs& a = ga[m];
                    a = mo.conv;
                    a.in = R.ptr(mo.in[0]);
                    a.out = R.ptr(mo.out[0]);
                    a.residual = mo.in.size() > 1 ? R.ptr(mo.in[1]) : nullptr;
                    a.up_in = nullptr;
                    a.wgt = W + mo.w_off;
                    a.bias = reinterpret_cast<const float*>(W + mo.b_off);
                    a.cscale = a.in_i8 ? reinterpret_cast<const float*>(W + mo.s_off) : nullptr;
                    a.N = nb(plan.tensors[mo.in[0]]);
                    a.M = a.N * a.Ho * a.Wo;
                }
                if (prof && probes[k].start && probes[k].stop) conv_set_launch_probe(&probes[k]);
                st = conv_igemm_group_f16(ga, gn, stream);
                conv_set_launch_probe(nullptr);
                if (st == TRTX_ERR_UNSUPPORTED) {   // a batch at which the members no longer share an instantiation: one launch each, same bits
                    st = TRTX_OK;
                    for (int m = 0; m < gn && st == TRTX_OK; ++m) st = conv_igemm_f16(ga[m], stream);
                }
                break;
            }
            case OP_POOL:
                st = nhwc_pool(R.ptr(op.in[0]), R.ptr(op.out[0]), op.dtype, op.i[0], nb(t0), t0.H, t0.W, t0.C, t0.ld, to.H,
                               to.W, to.ld, op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], op.i[7], stream);
                break;
            case OP_POOL_CHAIN: {
                const PTensor &y1 = plan.tensors[op.out[0]], &y2 = plan.tensors[op.out[1]], &y3 = plan.tensors[op.out[2]];
                st = nhwc_maxpool_chain3_f16(R.ptr(op.in[0]), R.ptr(op.out[0]), R.ptr(op.out[1]), R.ptr(op.out[2]), nb(t0), t0.H,
                                             t0.W, t0.C, t0.ld, y1.ld, y2.ld, y3.ld, op.i[1], stream, op.dtype == DT_F32 ? 1 : 0);
                break;
            }
            case OP_D2S:
                st = nhwc_depth_to_space_f16(R.ptr(op.in[0]), R.ptr(op.out[0]), nb(t0), t0.H, t0.W, to.C, op.i[0], op.i[1], t0.ld,
                                             to.ld, stream);
                break;
            case OP_RESIZE:
                if (t0.dtype == DT_I8) {  // int8 in place; requantised when the two owners were calibrated to different scales
                    st = nhwc_resize_nearest_i8(R.ptr(op.in[0]), R.ptr(op.out[0]), nb(t0), t0.H, t0.W, t0.C, t0.ld, to.H, to.W, to.ld,
                                                t0.scale / to.scale, stream);
                    break;
                }
                st = nhwc_resize_nearest(R.ptr(op.in[0]), R.ptr(op.out[0]), op.dtype, nb(t0), t0.H, t0.W, t0.C, t0.ld, to.H,
                                         to.W, to.ld, stream);
                break;
            case OP_EW_NHWC: {
                const PTensor& t1 = plan.tensors[op.in[1]];
                st = nhwc_elementwise(R.ptr(op.in[0]), R.ptr(op.in[1]), R.ptr(op.out[0]), op.dtype, op.i[0],
                                      (long)nb(t0) * t0.H * t0.W, t0.C, t0.ld, t1.ld, to.ld, stream);
                break;
            }
            case OP_ACT_NHWC:
                st = nhwc_activation(R.ptr(op.in[0]), R.ptr(op.out[0]), op.dtype, op.i[0], op.f[0],
                                     (long)nb(t0) * t0.H * t0.W, t0.C, t0.ld, to.ld, stream);
                break;
            case OP_SCALE_NHWC:
                st = nhwc_scale(R.ptr(op.in[0]), R.ptr(op.out[0]), op.dtype, reinterpret_cast<const float*>(W + op.s_off),
                                reinterpret_cast<const float*>(W + op.b_off), (long)nb(t0) * t0.H * t0.W, t0.C, t0.ld,
                                to.ld, stream);
                break;
            case OP_COPY_NHWC:
                st = nhwc_copy(R.ptr(op.in[0]), R.ptr(op.out[0]), op.dtype, (long)nb(t0) * t0.H * t0.W, t0.C, t0.ld, to.ld,
                               stream);
                break;
            case OP_REDUCE_HW:
                st = nhwc_reduce_hw_avg(R.ptr(op.in[0]), R.ptr(op.out[0]), op.dtype, nb(t0), t0.H * t0.W, t0.C, t0.ld, to.ld,
                                        stream);
                break;
            case OP_TO_NHWC:
                st = nchw_f32_to_nhwc(static_cast<const float*>(R.ptr(op.in[0])), R.ptr(op.out[0]), to.dtype, nb(to), to.C,
                                      to.H, to.W, to.Calloc ? to.Calloc : to.C, to.ld, stream);
                break;
            case OP_TO_LINEAR:
                st = nhwc_to_nchw_f32(R.ptr(op.in[0]), t0.dtype, static_cast<float*>(R.ptr(op.out[0])), nb(t0), t0.C, t0.H,
                                      t0.W, t0.ld, stream);
                break;
            case OP_GATHER: {
                const float* src = static_cast<const float*>(R.ptr(op.in[0])) + op.off0;
                const StridedView v = to.batched ? batch_view(op.view, batch, t0.batched ? t0.dims.volume() : 0, 0) : op.view;
                st = lin_gather(src, static_cast<float*>(R.ptr(op.out[0])), v, stream);
                break;
            }
            case OP_SCATTER: {
                float* dst = static_cast<float*>(R.ptr(op.out[0])) + op.off0;
                // dense source per sample; destination strided with the output's sample stride
                StridedView v = op.view;
                if (to.batched) {
                    if (!t0.batched) {
                        // broadcast an unbatched source into every sample: one launch per sample
                        for (int b = 0; b < batch && st == TRTX_OK; ++b)
                            st = lin_scatter(static_cast<const float*>(R.ptr(op.in[0])), dst + (long)b * to.dims.volume(), v,
                                             stream);
                        break;
                    }
                    v = batch_view(op.view, batch, to.dims.volume(), 0);
                }
                st = lin_scatter(static_cast<const float*>(R.ptr(op.in[0])), dst, v, stream);
                break;
            }
            case OP_EW_LIN: {
                const PTensor& t1 = plan.tensors[op.in[1]];
                const StridedView v = to.batched ? batch_view(op.view, batch, t0.batched ? t0.dims.volume() : 0,
                                                              t1.batched ? t1.dims.volume() : 0)
                                                 : op.view;
                st = lin_elementwise(static_cast<const float*>(R.ptr(op.in[0])), static_cast<const float*>(R.ptr(op.in[1])),
                                     static_cast<float*>(R.ptr(op.out[0])), op.i[0], v, stream);
                break;
            }
            case OP_ACT_LIN:
                st = lin_activation(static_cast<const float*>(R.ptr(op.in[0])), static_cast<float*>(R.ptr(op.out[0])), op.i[0],
                                    op.f[0], (long)(to.batched ? batch : 1) * to.dims.volume(), stream);
                break;
            case OP_SCALE_LIN: {
                const long bmul = to.batched ? batch : 1;
                st = lin_scale(static_cast<const float*>(R.ptr(op.in[0])), static_cast<float*>(R.ptr(op.out[0])),
                               reinterpret_cast<const float*>(W + op.s_off), reinterpret_cast<const float*>(W + op.b_off),
                               reinterpret_cast<const float*>(W + op.w_off), op.i[0], bmul * op.i[1], op.i[2], op.i[3],
                               stream);
                break;
            }
            case OP_SOFTMAX:
                st = lin_softmax(static_cast<const float*>(R.ptr(op.in[0])), static_cast<float*>(R.ptr(op.out[0])),
                                 (long)(to.batched ? batch : 1) * op.i[0], op.i[1], op.i[2], stream);
                break;
            case OP_MATMUL: {
                const PTensor& t1 = plan.tensors[op.in[1]];
                st = lin_matmul(static_cast<const float*>(R.ptr(op.in[0])), static_cast<const float*>(R.ptr(op.in[1])),
                                static_cast<float*>(R.ptr(op.out[0])), to.batched ? batch : 1, op.i[0], op.i[1], op.i[2],
                                op.i[3], op.i[4], t0.batched ? t0.dims.volume() : 0, t1.batched ? t1.dims.volume() : 0,
                                stream);
                break;
            }
            case OP_REDUCE_LIN:
                st = lin_reduce(static_cast<const float*>(R.ptr(op.in[0])), static_cast<float*>(R.ptr(op.out[0])), op.i[0],
                                (long)(to.batched ? batch : 1) * op.i[1], op.i[2], op.i[3], stream);
                break;
            case OP_PLUGIN: {
                std::vector<const void*> ins;
                std::vector<void*> outs;
                for (int t : op.in) ins.push_back(R.ptr(t));
                for (int t : op.out) outs.push_back(R.ptr(t));
                void* ws = op.ws_bytes ? static_cast<char*>(c->d_arena) + op.ws_off : nullptr;
                const int rc = op.plugin->v.enqueue(op.plugin->v.self, batch, ins.data(), outs.data(), ws, stream);
                if (rc != 0) {
                    fprintf(stderr, "[trtx_hip] plugin %s enqueue returned %d\n", op.name.c_str(), rc);
                    st = TRTX_ERR_HIP;
                }
                break;
            }
            case OP_YOLO_HEAD: {
                const int nl = op.i[4];
                const void* heads[8];
                int lds[8];
                for (int k = 0; k < nl; ++k) {
                    heads[k] = R.ptr(op.in[k]);
                    lds[k] = plan.tensors[op.in[k]].ld;
                }
                st = (t0.dtype == DT_F32 ? trtx_yolo_head_decode_nhwc_f32 : trtx_yolo_head_decode_nhwc)(
                        heads, lds, nl, batch, op.i[0], op.i[1], op.i[2], &op.i[5], reinterpret_cast<const float*>(W + op.w_off), op.i[3],
                        static_cast<float*>(R.ptr(op.out[0])), static_cast<char*>(c->d_arena) + op.ws_off, op.ws_bytes, stream);
                break;
            }
            case OP_ROI_ALIGN: {
                const PTensor& ft = plan.tensors[op.in[1]];
                st = trtx_roi_align_nhwc_f16_strided(batch, static_cast<const float*>(R.ptr(op.in[0])), R.ptr(op.in[1]), ft.ld, op.i[0], op.f[0], op.i[1],
                                                     op.i[2], ft.C, ft.H, ft.W, R.ptr(op.out[0]), to.ld, op.i[3] > 0 ? op.i[3] : 1, stream);
                break;
            }
            case OP_COPY_LIN: {
                const size_t bytes = (size_t)(to.batched ? batch : 1) * to.dims.volume() * 4;
                if (hipMemcpyAsync(R.ptr(op.out[0]), R.ptr(op.in[0]), bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                    st = TRTX_ERR_HIP;
                break;
            }
            default:
                st = TRTX_ERR_UNSUPPORTED;
        }
        if (st != TRTX_OK) {
            fprintf(stderr, "[trtx_hip] op %zu (%s, %s) failed: %s\n", k, op_kind_name(op.kind), op.name.c_str(),
                    trtx_status_string(st));
            return bail(st);
        }
        if (c->observer) {  // INT8 calibration: |x| maximum or histogram of every fp16 NHWC tensor this op wrote, per owning storage
            for (int t : op.out) {
                const PTensor& pt = plan.tensors[t];
                if (pt.layout != LAY_NHWC || pt.dtype != DT_F16 || pt.C % 8 || pt.ld % 8) continue;
                const long pixels = (long)nb(pt) * pt.H * pt.W;
                CalibObserver& ob = *c->observer;
                if (ob.mode == 1)
                    st = nhwc_absmax_f16(R.ptr(t), pixels, pt.C, pt.ld, ob.d_max + pt.storage, stream);
                else if (ob.mode == 2 && ob.range[pt.storage] > 0.f)
                    st = nhwc_hist_f16(R.ptr(t), pixels, pt.C, pt.ld, ob.range[pt.storage], ob.d_hist + (size_t)pt.storage * kCalibBins, stream);
                if (st != TRTX_OK) return bail(st);
            }
        }
        if (prof && hipEventRecord(evs[k + 1], stream) != hipSuccess) return bail(TRTX_ERR_HIP);
        if (lanes && op.signal && hipEventRecord(c->op_event[k], stream) != hipSuccess) return bail(TRTX_ERR_HIP);
    }
    if (lanes)
        for (int l = 1; l < plan.num_lanes; ++l)
            if (lane_started[l]) {  // join: the caller's stream continues after every lane has drained
                if (hipEventRecord(c->lane_done[l], c->lane_stream[l]) != hipSuccess || hipStreamWaitEvent(user_stream, c->lane_done[l], 0) != hipSuccess)
                    return bail(TRTX_ERR_HIP);
            }
    if (prof) {
        if (hipStreamSynchronize(stream) != hipSuccess) return bail(TRTX_ERR_HIP);
        for (size_t k = 0; k < plan.ops.size(); ++k) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, evs[k], evs[k + 1]);
            OpTiming t{plan.ops[k].name, op_kind_name(plan.ops[k].kind), ms};
            if (probes[k].launches == 1 && probes[k].start && probes[k].stop) {
                float kms = -1.f;
                if (hipEventElapsedTime(&kms, probes[k].start, probes[k].stop) == hipSuccess && kms > 0.f) t.kernel_ms = kms;
                else (void)hipGetLastError();
            }
            prof->push_back(t);
        }
        for (auto& ev : evs) (void)hipEventDestroy(ev);
        free_probes();
    }
    return TRTX_OK;
}

}  // namespace trtx

// ---------------------------------------------------------------------------------------------------------
struct trtx_hostmem {
    std::vector<uint8_t> data;
};

extern "C" const void* trtx_hostmem_data(const trtx_hostmem* m) { return m ? m->data.data() : nullptr; }
extern "C" size_t trtx_hostmem_size(const trtx_hostmem* m) { return m ? m->data.size() : 0; }
extern "C" void trtx_hostmem_destroy(trtx_hostmem* m) { delete m; }
trtx_hostmem* trtx_hostmem_from(std::vector<uint8_t>&& v) {
    auto* m = new trtx_hostmem();
    m->data = std::move(v);
    return m;
}

extern "C" int32_t trtx_hostmem_create(const void* data, size_t size, trtx_hostmem** out) {
    if (!out || (!data && size)) return TRTX_ERR_INVALID;
    const uint8_t* p = static_cast<const uint8_t*>(data);
    *out = trtx_hostmem_from(std::vector<uint8_t>(p, p + size));
    return TRTX_OK;
}

extern "C" void trtx_string_free(char* s) { free(s); }

extern "C" int32_t trtx_plan_describe(const void* plan_data, size_t size, int32_t lowered, char** json_out) {
    if (!plan_data || !json_out) return TRTX_ERR_INVALID;
    std::string err;
    auto net = Network::deserialize(static_cast<const uint8_t*>(plan_data), size, &err);
    if (!net) {
        fprintf(stderr, "[trtx_hip] trtx_plan_describe: %s\n", err.c_str());
        return TRTX_ERR_IO;
    }
    std::string js;
    if (!lowered) {
        js = net->describe_json();
    } else {
        Plan plan;
        if (!lower_network(*net, &plan)) {
            fprintf(stderr, "[trtx_hip] lowering failed: %s\n", plan.error.c_str());
            return TRTX_ERR_UNSUPPORTED;
        }
        pack_weights(*net, &plan);
        js = plan.describe_json();
    }
    *json_out = strdup(js.c_str());
    return TRTX_OK;
}

// An engine's weights, its contexts' arenas, lane streams and events all belong to ONE HIP device: the device that was current
// at deserializeCudaEngine (tutorials/multi_GPU_processing.md of the reference: cudaSetDevice(i) before building / using the i-th
// Plan).  Using it with another device current would launch kernels over foreign memory, so that is refused, loudly.
static int32_t check_device(const trtx_engine* e, const char* what) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return TRTX_ERR_HIP;
    }
    if (dev != e->device) {
        fprintf(stderr, "[trtx_hip] %s: the engine lives on HIP device %d but device %d is current - call hipSetDevice(%d) first\n", what,
                e->device, dev, e->device);
        return TRTX_ERR_STATE;
    }
    return TRTX_OK;
}

extern "C" int32_t trtx_engine_device(const trtx_engine* e) { return e ? e->device : -1; }

extern "C" int32_t trtx_engine_deserialize(const void* plan_data, size_t size, trtx_engine** out) {
    return trtx::engine_from_plan(plan_data, size, false, out);
}

int32_t trtx::engine_from_plan(const void* plan_data, size_t size, bool time_tactics, trtx_engine** out) {
    if (!plan_data || !out) return TRTX_ERR_INVALID;
    if (trtx_device_count() < 1) {
        fprintf(stderr, "[trtx_hip] no HIP device: engines only run on the GPU (there is no CPU fallback)\n");
        return TRTX_ERR_NO_DEVICE;
    }
    std::string err;
    std::unique_ptr<trtx_engine> e(new trtx_engine());
    TRTX_HIP_TRY(hipGetDevice(&e->device));
    e->net = Network::deserialize(static_cast<const uint8_t*>(plan_data), size, &err);
    if (!e->net) {
        fprintf(stderr, "[trtx_hip] deserializeCudaEngine: %s\n", err.c_str());
        return TRTX_ERR_IO;
    }
    e->blob.assign(static_cast<const uint8_t*>(plan_data), static_cast<const uint8_t*>(plan_data) + size);
    if (!lower_network(*e->net, &e->plan)) {
        fprintf(stderr, "[trtx_hip] lowering failed: %s\n", e->plan.error.c_str());
        return TRTX_ERR_UNSUPPORTED;
    }
    pack_weights(*e->net, &e->plan);
    if (e->plan.weight_bytes) {
        TRTX_HIP_TRY(hipMalloc(&e->d_weights, e->plan.weight_bytes));
        TRTX_HIP_TRY(hipMemcpy(e->d_weights, e->plan.weight_blob.data(), e->plan.weight_bytes, hipMemcpyHostToDevice));
    }
    e->plan.weight_blob.clear();
    e->plan.weight_blob.shrink_to_fit();
    for (auto& op : e->plan.ops) {
        if (op.kind != OP_PLUGIN) continue;
        if (op.plugin->v.initialize && op.plugin->v.initialize(op.plugin->v.self) != 0) {
            fprintf(stderr, "[trtx_hip] plugin %s: initialize() failed\n", op.name.c_str());
            return TRTX_ERR_UNSUPPORTED;  // ~trtx_engine terminates exactly the plugins counted so far
        }
        ++e->plugins_initialized;
    }
    // kernel tactics: the plan's own (chosen by timing when it was built on a GPU machine); a plan built without a GPU has none and
    // is timed here only on request (TRTX_TUNE=1) - by default it runs the static defaults, reproducibly
    if (!time_tactics && !e->net->tactics_timed) {
        time_tactics = read_options().tune == 1;
    }
    if (const int32_t st = tune_engine(e.get(), time_tactics)) return st;
    *out = e.release();
    return TRTX_OK;
}

extern "C" int32_t trtx_engine_serialize(const trtx_engine* e, trtx_hostmem** out) {
    if (!e || !out) return TRTX_ERR_INVALID;
    std::vector<uint8_t> copy = e->blob;
    *out = trtx_hostmem_from(std::move(copy));
    return TRTX_OK;
}

extern "C" void trtx_engine_destroy(trtx_engine* e) { delete e; }

extern "C" int32_t trtx_engine_nb_bindings(const trtx_engine* e) { return e ? (int32_t)e->plan.binding_tensor.size() : 0; }

extern "C" const char* trtx_engine_binding_name(const trtx_engine* e, int32_t i) {
    if (!e || i < 0 || i >= (int32_t)e->plan.binding_tensor.size()) return nullptr;
    return e->net->tensors[e->plan.binding_tensor[i]].name.c_str();
}

extern "C" int32_t trtx_engine_binding_index(const trtx_engine* e, const char* name) {
    if (!e || !name) return -1;
    for (size_t i = 0; i < e->plan.binding_tensor.size(); ++i)
        if (e->net->tensors[e->plan.binding_tensor[i]].name == name) return (int32_t)i;
    return -1;
}

extern "C" int32_t trtx_engine_binding_is_input(const trtx_engine* e, int32_t i) {
    if (!e || i < 0 || i >= (int32_t)e->plan.binding_tensor.size()) return 0;
    return e->plan.binding_is_input[i] ? 1 : 0;
}

extern "C" int32_t trtx_engine_binding_dims(const trtx_engine* e, int32_t i, trtx_dims* out) {
    if (!e || !out || i < 0 || i >= (int32_t)e->plan.binding_tensor.size()) return TRTX_ERR_INVALID;
    *out = to_c(e->net->tensors[e->plan.binding_tensor[i]].dims);
    return TRTX_OK;
}

extern "C" int32_t trtx_engine_binding_dtype(const trtx_engine*, int32_t) { return TRTX_DTYPE_FLOAT; }

extern "C" int32_t trtx_engine_max_batch(const trtx_engine* e) { return e ? e->plan.max_batch : 0; }

extern "C" size_t trtx_engine_device_memory(const trtx_engine* e) {
    return e ? e->plan.arena_bytes + e->plan.weight_bytes : 0;
}

extern "C" int32_t trtx_context_create(trtx_engine* e, trtx_context** out) {
    if (!e || !out) return TRTX_ERR_INVALID;
    if (const int32_t st = check_device(e, "createExecutionContext")) return st;
    std::unique_ptr<trtx_context> c(new trtx_context());
    c->engine = e;
    c->addr.assign(e->plan.binding_tensor.size(), nullptr);
    if (e->plan.arena_bytes) {
        TRTX_HIP_TRY(hipMalloc(&c->d_arena, e->plan.arena_bytes));
        TRTX_HIP_TRY(hipMemset(c->d_arena, 0, e->plan.arena_bytes));
    }
    const Plan& plan = e->plan;
    if (plan.num_lanes > 1) {
        c->lane_stream.assign(plan.num_lanes, nullptr);
        c->lane_done.assign(plan.num_lanes, nullptr);
        c->op_event.assign(plan.ops.size(), nullptr);
        TRTX_HIP_TRY(hipEventCreateWithFlags(&c->start_event, hipEventDisableTiming));
        for (int l = 1; l < plan.num_lanes; ++l) {
            TRTX_HIP_TRY(hipStreamCreateWithFlags(&c->lane_stream[l], hipStreamNonBlocking));
            TRTX_HIP_TRY(hipEventCreateWithFlags(&c->lane_done[l], hipEventDisableTiming));
        }
        for (size_t k = 0; k < plan.ops.size(); ++k)
            if (plan.ops[k].signal) TRTX_HIP_TRY(hipEventCreateWithFlags(&c->op_event[k], hipEventDisableTiming));
    }
    *out = c.release();
    return TRTX_OK;
}

extern "C" void trtx_context_destroy(trtx_context* c) { delete c; }

namespace {

// The lowered plan is a fixed launch sequence (60-90 kernels over up to 6 lanes with event fences) that depends only on
// (batch, binding pointers): the first enqueue of a combination is captured into a hipGraph through the caller's stream
// (fork/join across the lane streams is expressed by the same events) and later enqueues replay it with one
// hipGraphLaunch: per-launch host work and the inter-kernel dispatch gaps of 60-90 separate launches go away.
// Not captured: plans that call user plugins (they may synchronise / allocate), the NULL stream.
// Measured on YOLOv8n b32 (profiles/r02_graph_vs_eager.txt): 1.572 ms/step replayed vs 1.564 ms eager - the step is bound by
// the dependent chain of ~60 short kernels on the GPU, not by host launch cost or dispatch gaps - so replay is OPT-IN
// (TRTX_GRAPH=1) and the default stays the eager multi-stream executor.
int32_t enqueue_maybe_graph(trtx_context* c, int batch, void* const* bindings, hipStream_t stream) {
    const trtx::Plan& plan = c->engine->plan;
    if (c->graph_state == 0) {
        static const bool on = options().graph;
        bool ok = on;
        for (const auto& op : plan.ops)
            if (op.kind == trtx::OP_PLUGIN && !trtx::builtin_plugin_capturable(op.plugin->v)) ok = false;
        c->graph_state = ok ? 1 : -1;
    }
    ++c->enqueue_count;
    if (c->graph_state < 0 || stream == nullptr) return trtx::execute_plan(c, batch, bindings, stream, nullptr);
    const size_t nb = plan.binding_tensor.size();
    for (auto& g : c->graphs) {
        if (g.batch != batch || memcmp(g.bindings.data(), bindings, nb * sizeof(void*)) != 0) continue;
        g.last_use = c->enqueue_count;
        TRTX_HIP_TRY(hipGraphLaunch(g.exec, stream));
        return TRTX_OK;
    }
    // capture on the SECOND sighting of a (batch, pointers) combination: the first run is eager (it also performs every lazy
    // one-time initialisation outside a capture), and a caller that passes fresh buffers on every call never pays for captures
    bool seen_before = false;
    for (const auto& sn : c->seen)
        if (sn.batch == batch && memcmp(sn.bindings.data(), bindings, nb * sizeof(void*)) == 0) seen_before = true;
    if (!seen_before) {
        if (c->seen.size() >= 32) c->seen.erase(c->seen.begin());
        trtx_context::CapturedGraph sn;
        sn.batch = batch;
        sn.bindings.assign(bindings, bindings + nb);
        c->seen.push_back(sn);
        return trtx::execute_plan(c, batch, bindings, stream, nullptr);
    }
    for (size_t b = 0; b < nb; ++b)
        if (!bindings[b]) return TRTX_ERR_INVALID;
    trtx_context::CapturedGraph g;
    g.batch = batch;
    g.bindings.assign(bindings, bindings + nb);
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        c->graph_state = -1;
        return trtx::execute_plan(c, batch, bindings, stream, nullptr);
    }
    const int32_t st = trtx::execute_plan(c, batch, bindings, stream, nullptr);
    const hipError_t e = hipStreamEndCapture(stream, &g.graph);
    if (st != TRTX_OK || e != hipSuccess || !g.graph || hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (g.graph) (void)hipGraphDestroy(g.graph);
        c->graph_state = -1;  // fall back to eager launches for good
        fprintf(stderr, "[trtx_hip] hipGraph capture of the plan failed (%s): eager launches from now on\n", hipGetErrorString(e));
        return st != TRTX_OK ? st : trtx::execute_plan(c, batch, bindings, stream, nullptr);
    }
    if (c->graphs.size() >= 16) {  // bounded cache: drop the least recently used graph
        size_t lru = 0;
        for (size_t i = 1; i < c->graphs.size(); ++i)
            if (c->graphs[i].last_use < c->graphs[lru].last_use) lru = i;
        (void)hipGraphExecDestroy(c->graphs[lru].exec);
        (void)hipGraphDestroy(c->graphs[lru].graph);
        c->graphs.erase(c->graphs.begin() + lru);
    }
    g.last_use = c->enqueue_count;
    c->graphs.push_back(g);
    TRTX_HIP_TRY(hipGraphLaunch(g.exec, stream));
    return TRTX_OK;
}

}  // namespace

extern "C" int32_t trtx_context_enqueue(trtx_context* c, int32_t batch, void* const* bindings, trtx_stream_t stream) {
    if (!c || !bindings) return TRTX_ERR_INVALID;
    const int b = c->engine->plan.explicit_batch ? 1 : batch;
    if (b < 1 || b > c->engine->plan.max_batch) {
        fprintf(stderr, "[trtx_hip] enqueue: batch %d outside [1, %d]\n", b, c->engine->plan.max_batch);
        return TRTX_ERR_INVALID;
    }
    if (const int32_t st = check_device(c->engine, "enqueue")) return st;
    return enqueue_maybe_graph(c, b, bindings, stream);
}

// Camera frames in, detections out: the stem convolution samples the letterboxed frame itself (kernels/conv_stem.hip FRAMES), so the
// fp32 network-input tensor of cuda_batch_preprocess + enqueue (yolov8_det.cpp:146-160) never exists.  Eager path only (no graph replay:
// the frame pointers are kernel arguments).
extern "C" int32_t trtx_context_enqueue_frames(trtx_context* c, int32_t batch, const void* const* frames, const int32_t* frame_w, const int32_t* frame_h,
                                               void* const* bindings, trtx_stream_t stream) {
    if (!c || !bindings || !frames || !frame_w || !frame_h) return TRTX_ERR_INVALID;
    const Plan& plan = c->engine->plan;
    if (plan.explicit_batch || batch < 1 || batch > plan.max_batch) return TRTX_ERR_INVALID;
    if (const int32_t st = check_device(c->engine, "enqueue_frames")) return st;
    // the plan must start with a 3-channel stem convolution reading the (only) input binding
    int stem = -1, n_in = 0;
    for (size_t b = 0; b < plan.binding_is_input.size(); ++b) n_in += plan.binding_is_input[b] ? 1 : 0;
    for (size_t k = 0; k < plan.ops.size() && stem < 0; ++k)
        if (plan.ops[k].kind == OP_CONV && plan.ops[k].stem) stem = (int)k;
    if (n_in != 1 || stem < 0 || plan.ops[stem].conv.Cin != 3) return TRTX_ERR_UNSUPPORTED;
    if (plan.ops[stem].conv.f32) return TRTX_ERR_UNSUPPORTED;   // fp32 engines have a stem op too (round 5) but no fused-letterbox form of it: rejected before anything is enqueued
    for (size_t k = 0; k < plan.ops.size(); ++k)   // nothing else may read the input tensor
        for (int t : plan.ops[k].in)
            if ((int)k != stem && t == plan.ops[stem].in[0]) return TRTX_ERR_UNSUPPORTED;
    const ConvArgs& a = plan.ops[stem].conv;
    std::vector<StemFrame> fr(batch);
    for (int i = 0; i < batch; ++i) {
        if (!frames[i] || frame_w[i] < 1 || frame_h[i] < 1) return TRTX_ERR_INVALID;
        fr[i].src = frames[i];
        fr[i].w = frame_w[i];
        fr[i].h = frame_h[i];
        trtx_letterbox_matrix(frame_w[i], frame_h[i], a.W, a.H, fr[i].d2s);
    }
    c->frames = fr.data();
    const int32_t st = execute_plan(c, batch, bindings, stream, nullptr);
    c->frames = nullptr;
    return st;
}

extern "C" int32_t trtx_context_set_tensor_address(trtx_context* c, const char* name, void* ptr) {
    if (!c || !name) return TRTX_ERR_INVALID;
    const int i = trtx_engine_binding_index(c->engine, name);
    if (i < 0) return TRTX_ERR_INVALID;
    c->addr[i] = ptr;
    return TRTX_OK;
}

extern "C" int32_t trtx_context_enqueue_v3(trtx_context* c, trtx_stream_t stream) {
    if (!c) return TRTX_ERR_INVALID;
    for (void* p : c->addr)
        if (!p) return TRTX_ERR_STATE;
    if (const int32_t st = check_device(c->engine, "enqueueV3")) return st;
    return enqueue_maybe_graph(c, c->engine->plan.explicit_batch ? 1 : c->engine->plan.max_batch, c->addr.data(), stream);
}

extern "C" int32_t trtx_context_profile(trtx_context* c, int32_t batch, void* const* bindings, trtx_stream_t stream,
                                        char** json_out) {
    if (!c || !bindings || !json_out) return TRTX_ERR_INVALID;
    if (const int32_t st = check_device(c->engine, "profile")) return st;
    std::vector<OpTiming> prof;
    const int32_t st = execute_plan(c, c->engine->plan.explicit_batch ? 1 : batch, bindings, stream, &prof);
    if (st != TRTX_OK) return st;
    std::ostringstream o;
    o << "[";
    for (size_t k = 0; k < prof.size(); ++k) {
        o << (k ? "," : "") << "{\"name\":\"";
        for (char ch : prof[k].name) o << ((ch == '"' || ch == '\\' || (unsigned char)ch < 0x20) ? ' ' : ch);
        o << "\",\"kind\":\"" << prof[k].kind << "\",\"ms\":" << prof[k].ms << ",\"kernel_ms\":" << prof[k].kernel_ms << "}";
    }
    o << "]";
    *json_out = strdup(o.str().c_str());
    return TRTX_OK;
}
