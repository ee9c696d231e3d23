#include "int8.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <memory>

#include "../common.h"
#include "../options.h"
#include "engine.h"

namespace trtx {

float clip_limited_threshold(const std::vector<double>& hist, float range, float thr, double limit) {
    const int n = (int)hist.size();
    if (!(limit > 0.0) || n < 1 || !(range > 0.f)) return thr;
    double total = 0;
    for (double v : hist) total += v;
    if (!(total > 0)) return thr;
    const double w = (double)range / n;
    // walk down from the top: bins k .. n - 1 lie beyond edge k * w (their centres do); the lowest edge that keeps their share within the limit
    double beyond = 0;
    int k = n;
    while (k > 0 && (beyond + hist[k - 1]) <= limit * total) beyond += hist[--k];
    const float edge = (float)(k * w);
    return edge > thr ? edge : thr;
}

float entropy_threshold(const std::vector<double>& hist_in, float range) {
    // Bin 0 takes the value of bin 1 before the search, as NVIDIA's public restatement of this calibrator does (pytorch-quantization,
    // calib/histogram.py::_compute_amax_entropy: "bins[0] = bins[1]"): the exact zeros of a post-ReLU tensor (and zero padding) are
    // representable at ANY scale, but as a spike in bin 0 they dominate the divergence - level 0 of the candidate averages the spike
    // over i/128 source bins, a term that grows with the threshold - and drag it down to ~1.9 sigma on ReLU(N(0,1)), clipping 3 % of the
    // tensor (round 3's RetinaFace int8 engines lost half their detections to exactly this; with the fix the same data gives 4.7 sigma).
    std::vector<double> hist(hist_in);
    if (hist.size() > 1) hist[0] = hist[1];
    const int nbins = (int)hist.size();
    const int levels = 128;
    double total = 0;
    for (double h : hist) total += h;
    if (!(total > 0) || !(range > 0.f)) return range;
    double best_kl = 1e300;
    int best_i = nbins;
    std::vector<double> p, q;
    for (int i = levels; i <= nbins; ++i) {
        // reference distribution: the first i bins, everything beyond clipped into the last of them
        p.assign(hist.begin(), hist.begin() + i);
        double outliers = 0;
        for (int k = i; k < nbins; ++k) outliers += hist[k];
        p[i - 1] += outliers;
        // candidate: the first i bins merged into 128 levels, each level spread back evenly over its non-empty source bins
        q.assign(i, 0.0);
        const int merged = i / levels;
        for (int j = 0; j < levels; ++j) {
            const int start = j * merged, stop = (j == levels - 1) ? i : start + merged;
            double sum = 0;
            int nz = 0;
            for (int k = start; k < stop; ++k) {
                sum += hist[k];
                nz += hist[k] > 0 ? 1 : 0;
            }
            if (!nz) continue;
            const double v = sum / nz;
            for (int k = start; k < stop; ++k)
                if (hist[k] > 0) q[k] = v;
        }
        double sp = 0, sq = 0;
        for (int k = 0; k < i; ++k) {
            sp += p[k];
            sq += q[k];
        }
        if (!(sp > 0) || !(sq > 0)) continue;
        double kl = 0;
        for (int k = 0; k < i; ++k) {
            if (!(p[k] > 0)) continue;
            const double pk = p[k] / sp;
            const double qk = q[k] > 0 ? q[k] / sq : 1e-12;  // mass the quantised distribution cannot represent
            kl += pk * log(pk / qk);
        }
        if (kl < best_kl) {
            best_kl = kl;
            best_i = i;
        }
    }
    return ((float)best_i + 0.5f) * (range / (float)nbins);
}

std::string calib_tensor_name(const Network& net, int t) {
    if (t >= 0 && t < (int)net.tensors.size() && !net.tensors[t].name.empty()) return net.tensors[t].name;
    return "(Unnamed Tensor* " + std::to_string(t) + ")";
}

std::string write_calibration_cache(const Network& net) {
    std::string out = "TRT-8601-EntropyCalibration2\n";
    for (size_t t = 0; t < net.tensor_scale.size() && t < net.tensors.size(); ++t) {
        const float sc = net.tensor_scale[t];
        if (!(sc > 0.f)) continue;
        uint32_t bits;
        memcpy(&bits, &sc, 4);
        char hex[16];
        snprintf(hex, sizeof hex, "%08x", bits);
        out += calib_tensor_name(net, (int)t) + ": " + hex + "\n";
    }
    return out;
}

bool read_calibration_cache(const void* data, size_t length, Network* net, std::string* err) {
    const std::string text(static_cast<const char*>(data), length);
    size_t pos = text.find('\n');
    if (pos == std::string::npos || text.compare(0, 4, "TRT-") != 0 || text.substr(0, pos).find("Calibration") == std::string::npos) {
        if (err) *err = "calibration cache: missing 'TRT-...-EntropyCalibration2' header";
        return false;
    }
    std::map<std::string, float> by_name;
    ++pos;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        const std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        const size_t colon = line.rfind(": ");
        if (colon == std::string::npos) continue;
        const unsigned long bits = strtoul(line.c_str() + colon + 2, nullptr, 16);
        const uint32_t b32 = (uint32_t)bits;
        float sc;
        memcpy(&sc, &b32, 4);
        if (sc > 0.f && sc < 1e30f) by_name[line.substr(0, colon)] = sc;
    }
    net->tensor_scale.assign(net->tensors.size(), 0.f);
    size_t hits = 0;
    for (size_t t = 0; t < net->tensors.size(); ++t) {
        auto it = by_name.find(calib_tensor_name(*net, (int)t));
        if (it != by_name.end()) {
            net->tensor_scale[t] = it->second;
            ++hits;
        }
    }
    if (!hits) {
        if (err) *err = "calibration cache: no entry matches a tensor of this network";
        return false;
    }
    return true;
}

int32_t run_int8_calibration(Network* net, const trtx_calibrator_vtbl& calib) {
    if (!calib.get_batch || !calib.get_batch_size) return TRTX_ERR_INVALID;
    const bool minmax = calib.get_algorithm && calib.get_algorithm(calib.self) == 3;   // nvinfer1::CalibrationAlgoType::kMINMAX_CALIBRATION
    if (trtx_device_count() < 1) {
        fprintf(stderr, "[trtx_hip] INT8 calibration runs the network on the GPU: no HIP device (provide a calibration cache to build without one)\n");
        return TRTX_ERR_NO_DEVICE;
    }
    // the statistics come from an fp16 engine of the same definition
    Network fp16net = *net;
    fp16net.int8 = false;
    fp16net.fp16 = true;
    fp16net.tensor_scale.clear();
    const int batch = std::max(1, std::min(calib.get_batch_size(calib.self), net->explicit_batch ? 1 : 1 << 20));
    fp16net.max_batch = net->explicit_batch ? net->max_batch : batch;
    std::vector<uint8_t> blob;
    fp16net.serialize(blob);
    trtx_engine* eng = nullptr;
    int32_t st;
    {
        CalibrationLowering keep_every_tensor;   // no upsample fold, no fused chains: see plan.h
        st = trtx_engine_deserialize(blob.data(), blob.size(), &eng);
    }
    if (st != TRTX_OK) return st;
    trtx_context* ctx = nullptr;
    st = trtx_context_create(eng, &ctx);
    if (st != TRTX_OK) {
        trtx_engine_destroy(eng);
        return st;
    }
    const Plan& plan = eng->plan;
    const size_t ns = plan.storages.size();
    CalibObserver obs;
    obs.range.assign(ns, 0.f);
    std::vector<std::vector<double>> hist(ns);
    std::vector<void*> bindings(plan.binding_tensor.size(), nullptr);
    std::vector<void*> owned;
    std::vector<const char*> in_names;
    std::vector<int> in_slots;
    hipStream_t stream = nullptr;
    auto cleanup = [&]() {
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
            stream = nullptr;
        }
        for (void* p : owned) (void)hipFree(p);
        if (obs.d_max) (void)hipFree(obs.d_max);
        if (obs.d_hist) (void)hipFree(obs.d_hist);
        trtx_context_destroy(ctx);
        trtx_engine_destroy(eng);
    };
#define CAL_TRY(expr)                                                          \
    do {                                                                       \
        if ((expr) != hipSuccess) {                                            \
            fprintf(stderr, "[trtx_hip] INT8 calibration: %s failed\n", #expr); \
            cleanup();                                                         \
            return TRTX_ERR_HIP;                                               \
        }                                                                      \
    } while (0)
    CAL_TRY(hipMalloc(reinterpret_cast<void**>(&obs.d_max), ns * sizeof(unsigned)));
    CAL_TRY(hipMalloc(reinterpret_cast<void**>(&obs.d_hist), ns * kCalibBins * sizeof(unsigned long long)));
    CAL_TRY(hipMemset(obs.d_hist, 0, ns * kCalibBins * sizeof(unsigned long long)));
    for (size_t b = 0; b < plan.binding_tensor.size(); ++b) {
        const TensorDef& t = eng->net->tensors[plan.binding_tensor[b]];
        if (plan.binding_is_input[b]) {
            in_names.push_back(t.name.c_str());
            in_slots.push_back((int)b);
        } else {
            void* p = nullptr;
            CAL_TRY(hipMalloc(&p, (size_t)std::max<int64_t>(1, t.dims.volume()) * (size_t)fp16net.max_batch * 4));
            owned.push_back(p);
            bindings[b] = p;
        }
    }
    CAL_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    std::vector<void*> in_ptrs(in_names.size(), nullptr);
    std::vector<unsigned> h_max(ns);
    std::vector<unsigned long long> h_hist(ns * kCalibBins);
    int n_batches = 0;
    ctx->observer = &obs;
    while (calib.get_batch(calib.self, in_ptrs.data(), in_names.data(), (int32_t)in_names.size())) {
        bool ok = true;
        for (size_t i = 0; i < in_slots.size(); ++i) {
            bindings[in_slots[i]] = in_ptrs[i];
            ok = ok && in_ptrs[i] != nullptr;
        }
        if (!ok) {
            fprintf(stderr, "[trtx_hip] INT8 calibration: getBatch left an input binding null\n");
            st = TRTX_ERR_INVALID;
            break;
        }
        const int run_batch = net->explicit_batch ? 1 : batch;
        // pass 1: |x| maxima of this batch
        obs.mode = 1;
        CAL_TRY(hipMemsetAsync(obs.d_max, 0, ns * sizeof(unsigned), stream));
        st = execute_plan(ctx, run_batch, bindings.data(), stream, nullptr);
        if (st != TRTX_OK) break;
        CAL_TRY(hipMemcpyAsync(h_max.data(), obs.d_max, ns * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        CAL_TRY(hipStreamSynchronize(stream));
        // ranges only grow, by powers of two, and the histogram collected so far is re-binned on the device copy's host mirror
        bool rebinned = false;
        for (size_t s = 0; s < ns; ++s) {
            float m;
            memcpy(&m, &h_max[s], 4);
            if (!(m > 0.f)) continue;
            if (obs.range[s] == 0.f) {
                obs.range[s] = m;
                continue;
            }
            while (m > obs.range[s]) {
                if (!rebinned) {
                    CAL_TRY(hipMemcpy(h_hist.data(), obs.d_hist, h_hist.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                    rebinned = true;
                }
                unsigned long long* h = h_hist.data() + s * kCalibBins;
                for (int k = 0; k < kCalibBins / 2; ++k) h[k] = h[2 * k] + h[2 * k + 1];
                for (int k = kCalibBins / 2; k < kCalibBins; ++k) h[k] = 0;
                obs.range[s] *= 2.f;
            }
        }
        if (rebinned) CAL_TRY(hipMemcpy(obs.d_hist, h_hist.data(), h_hist.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        // pass 2: histograms over the (possibly widened) ranges
        obs.mode = 2;
        st = execute_plan(ctx, run_batch, bindings.data(), stream, nullptr);
        if (st != TRTX_OK) break;
        CAL_TRY(hipStreamSynchronize(stream));
        ++n_batches;
    }
    ctx->observer = nullptr;
    if (st == TRTX_OK && n_batches == 0) {
        fprintf(stderr, "[trtx_hip] INT8 calibration: the calibrator produced no batch\n");
        st = TRTX_ERR_INVALID;
    }
    if (st == TRTX_OK) {
        CAL_TRY(hipMemcpy(h_hist.data(), obs.d_hist, h_hist.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        net->tensor_scale.assign(net->tensors.size(), 0.f);
        // TRTX_CALIB_REPORT=<file>: per calibrated tensor the largest |x| seen, the threshold chosen, the share of elements beyond it
        // (what the int8 engine clips) and the share in bin 0 - the evidence behind a detection-level int8 figure
        FILE* report = nullptr;
        if (const std::string rp = read_options().calib_report; !rp.empty()) report = fopen(rp.c_str(), "a");
        if (report) fprintf(report, "# %s calibration, %d batch(es) of %d; tensor\tabsmax\tthreshold\tclipped_share\tbin0_share\telements\n", minmax ? "min-max" : "entropy", n_batches, batch);
        for (const PTensor& t : plan.tensors) {
            if (t.parent >= 0 || t.layout != LAY_NHWC || t.net_tensor < 0 || t.storage < 0) continue;
            const size_t s = (size_t)t.storage;
            if (!(obs.range[s] > 0.f)) continue;
            // the collected fine histogram -> kEntropyBins bins over [0, absmax], absmax = upper edge of the last occupied fine bin; a fine
            // bin that straddles two coarse ones is split in proportion (>= 2 fine bins per coarse bin: at most half a fine bin of smear)
            const unsigned long long* hf = h_hist.data() + s * kCalibBins;
            int last = 0;
            for (int k = 0; k < kCalibBins; ++k)
                if (hf[k]) last = k;
            const double wf = (double)obs.range[s] / kCalibBins;
            const float absmax = (float)((last + 1) * wf);
            std::vector<double> h(kEntropyBins, 0.0);
            const double wc = (double)absmax / kEntropyBins;
            for (int k = 0; k <= last; ++k) {
                if (!hf[k]) continue;
                const double lo = k * wf, hi = (k + 1) * wf;
                int c0 = (int)(lo / wc), c1 = (int)(hi / wc);
                c0 = c0 < kEntropyBins ? c0 : kEntropyBins - 1;
                c1 = c1 < kEntropyBins ? c1 : kEntropyBins - 1;
                if (c0 == c1) {
                    h[c0] += (double)hf[k];
                } else {
                    for (int c = c0; c <= c1; ++c) {
                        const double a = std::max(lo, c * wc), b = std::min(hi, (c + 1) * wc);
                        if (b > a) h[c] += (double)hf[k] * (b - a) / (hi - lo);
                    }
                }
            }
            float thr;
            if (minmax) {   // kMINMAX_CALIBRATION: the largest |x| seen (to one fine bin)
                thr = absmax;
            } else {
                // ... but never so low that more than int8_clip_limit of the tensor saturates: the KL search weighs the histogram's BULK, and on heavy-tailed
                // activations (SiLU networks) it trades away a tail of 1e-3 of the elements - the detection candidates live there
                thr = clip_limited_threshold(h, absmax, entropy_threshold(h, absmax), read_options().int8_clip_limit);
            }
            net->tensor_scale[t.net_tensor] = thr / 127.0f;
            if (report) {
                double total = 0, beyond = 0;
                for (int k = 0; k < kEntropyBins; ++k) {
                    total += h[k];
                    if ((k + 0.5) * wc > thr) beyond += h[k];
                }
                fprintf(report, "%s\t%.6g\t%.6g\t%.3e\t%.4f\t%.0f\n", calib_tensor_name(*net, t.net_tensor).c_str(), (double)absmax, (double)thr,
                        total > 0 ? beyond / total : 0.0, total > 0 ? h[0] / total : 0.0, total);
            }
        }
        if (report) fclose(report);
    }
    cleanup();
#undef CAL_TRY
    return st;
}

}  // namespace trtx
