// RetinaFace and R-CNN (detectron2-style) detection plugins for gfx950 — HIP re-implementations of
//   retinaface/decode.cu:110-191 (DecodePlugin) + retinaface/common.hpp:91-130 (host nms)
//   rcnn/RpnDecode.cu:27-143, RpnNms.cu:27-121, RoiAlign.cu:29-182, PredictorDecode.cu:24-110, BatchedNms.cu:28-162
// behind the C ABI of include/trtx_hip.h.
//
// Building blocks (all wave64, no atomically-ordered output anywhere => bit-reproducible):
//   * stable descending sort = ascending bitonic sort of 64-bit keys (~ord(score) << 32 | index):
//     2048-key tiles are sorted / merged in LDS, larger strides by global compare-exchange launches;
//   * exact greedy NMS (hard, soft-linear, soft-gaussian; optional class awareness): sorted boxes are resolved
//     64 at a time inside one wave (shuffle broadcast of the candidate suppressor), the surviving 64 are then
//     applied by the whole workgroup to all later boxes — n/64 barriers instead of the reference's n;
//   * ordered compaction by ballot/popcount scans.
// The arithmetic of every decode / IoU follows the reference operation by operation (this file is compiled with
// -ffp-contract=off; double-precision literals of the reference are kept double).
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"

namespace {

using trtx::align_up;
using trtx::ord_f32;

constexpr int kTile = 2048;     // keys per LDS tile
constexpr int kSortThreads = 1024;

inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------------------------- sort
__device__ __forceinline__ void cmpx(uint64_t& a, uint64_t& b, bool up) {
    if ((a > b) == up) {
        const uint64_t t = a;
        a = b;
        b = t;
    }
}

// stages k = 2 .. k_max (k_max <= kTile) entirely inside LDS; direction from the global index
__global__ __launch_bounds__(kSortThreads) void bitonic_tile_sort(uint64_t* __restrict__ keys, int n_pad, int k_max) {
    __shared__ uint64_t s[kTile];
    uint64_t* base = keys + (size_t)blockIdx.y * n_pad + (size_t)blockIdx.x * kTile;
    const int gbase = blockIdx.x * kTile;
    const int t = threadIdx.x;
    const int span = n_pad < kTile ? n_pad : kTile;
    for (int i = t; i < span; i += kSortThreads) s[i] = base[i];
    __syncthreads();
    for (int k = 2; k <= k_max; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (t < span / 2) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const bool up = ((gbase + i) & k) == 0;
                uint64_t a = s[i], b = s[i | j];
                cmpx(a, b, up);
                s[i] = a;
                s[i | j] = b;
            }
            __syncthreads();
        }
    }
    for (int i = t; i < span; i += kSortThreads) base[i] = s[i];
}

// for one global k (> kTile): strides j = kTile/2 .. 1 inside LDS
__global__ __launch_bounds__(kSortThreads) void bitonic_tile_merge(uint64_t* __restrict__ keys, int n_pad, int k) {
    __shared__ uint64_t s[kTile];
    uint64_t* base = keys + (size_t)blockIdx.y * n_pad + (size_t)blockIdx.x * kTile;
    const int gbase = blockIdx.x * kTile;
    const int t = threadIdx.x;
    for (int i = t; i < kTile; i += kSortThreads) s[i] = base[i];
    __syncthreads();
    const bool up = (gbase & k) == 0;
    for (int j = kTile >> 1; j > 0; j >>= 1) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        uint64_t a = s[i], b = s[i | j];
        cmpx(a, b, up);
        s[i] = a;
        s[i | j] = b;
        __syncthreads();
    }
    for (int i = t; i < kTile; i += kSortThreads) base[i] = s[i];
}

__global__ void bitonic_global_step(uint64_t* __restrict__ keys, int n_pad, int k, int j) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pad / 2) return;
    uint64_t* base = keys + (size_t)blockIdx.y * n_pad;
    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const bool up = (i & k) == 0;
    uint64_t a = base[i], b = base[i | j];
    if ((a > b) == up) {
        base[i] = b;
        base[i | j] = a;
    }
}

// ascending sort of [batch][n_pad] 64-bit keys, n_pad a power of two
int32_t sort_keys(uint64_t* keys, int batch, int n_pad, hipStream_t s) {
    const int tiles = n_pad > kTile ? n_pad / kTile : 1;
    hipLaunchKernelGGL(bitonic_tile_sort, dim3(tiles, batch), dim3(kSortThreads), 0, s, keys, n_pad,
                       n_pad < kTile ? n_pad : kTile);
    for (int k = kTile * 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j >= kTile; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_step, dim3((n_pad / 2 + 255) / 256, batch), dim3(256), 0, s, keys, n_pad, k, j);
        hipLaunchKernelGGL(bitonic_tile_merge, dim3(tiles, batch), dim3(kSortThreads), 0, s, keys, n_pad, k);
    }
    return trtx::check_launch("sort_keys");
}

// ---------------------------------------------------------------------------------------------- top-k selection
// RpnDecode keeps the top_n = 6000 of 63 000 objectness logits (rcnn/RpnDecode.cu:71-88 sorts all of them with cub).  Here one
// workgroup per image finds the top_n-th key with a 3-pass MSD radix select over the 32-bit order-preserving score keys
// (11 + 11 + 10 bits, histograms in LDS, the scores are read three times from L2), then compacts the selected elements IN INDEX
// ORDER (ties of the pivot score go to the lowest indices, which is what the stable descending sort would keep), emitting the
// same 64-bit (key, index) words as make_keys_kernel.  Only those <= top_n keys (padded to a power of two) are sorted.
// Round 4 (284 us -> see profiles/r04_*): the bin walk that finds the pivot digit is a parallel scan (was thread 0 over 2048 bins, ~17 us a
// pass), the histogram passes read four scores per lane and iteration, and the compaction has no workgroup barrier inside its loop: every
// wave owns a CONTIGUOUS index range, counts its selected / tied elements, one barrier turns the 16 counts into bases, then each wave writes
// its range with ballot / popcount offsets of its own.  Same selection, same order (index order, ties of the pivot to the lowest indices).
__global__ __launch_bounds__(1024) void topk_select_kernel(const float* __restrict__ scores, int n, int top_n, int n_sel_pad,
                                                           uint64_t* __restrict__ sel_keys) {
    __shared__ unsigned s_hist[2048];
    __shared__ unsigned s_wsum[16];
    __shared__ unsigned s_prefix, s_remaining;
    __shared__ int s_wsel[16], s_weq[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sc = scores + (size_t)b * n;
    if (tid == 0) {
        s_prefix = 0;
        s_remaining = (unsigned)top_n;
    }
    const int shifts[3] = {21, 10, 0};
    const int bits[3] = {11, 11, 10};
    const bool vec = (((uintptr_t)sc) & 15) == 0;
    const int n4 = vec ? n >> 2 : 0;
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const int hi_shift = pass == 0 ? 32 : shifts[pass - 1];
        const unsigned dmask = (1u << bits[pass]) - 1u;
        auto count = [&](float v) {
            const unsigned key = ~ord_f32(v);
            if (pass > 0 && (key >> hi_shift) != prefix) return;
            atomicAdd(&s_hist[(key >> shifts[pass]) & dmask], 1u);
        };
        for (int i = tid; i < n4; i += 1024) {
            const float4 v = reinterpret_cast<const float4*>(sc)[i];
            count(v.x); count(v.y); count(v.z); count(v.w);
        }
        for (int i = n4 * 4 + tid; i < n; i += 1024) count(sc[i]);
        __syncthreads();
        // the digit: the first bin at which the running count reaches `remaining`.  Thread t owns bins 2t, 2t + 1.
        {
            const unsigned rem = s_remaining;
            const unsigned h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
            unsigned incl = h0 + h1;                      // inclusive scan over the threads' pair sums: wave, then across waves
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            unsigned before = 0;
            for (int w = 0; w < wave; ++w) before += s_wsum[w];
            const unsigned excl = before + incl - (h0 + h1);   // count in the bins below 2t
            // exactly one bin satisfies excl_bin < rem <= excl_bin + hist[bin] (rem >= 1, the total count >= rem)
            int bin = -1;
            unsigned cum = 0;
            if (excl < rem && rem <= excl + h0) { bin = 2 * tid; cum = excl; }
            else if (excl + h0 < rem && rem <= excl + h0 + h1) { bin = 2 * tid + 1; cum = excl + h0; }
            if (bin >= 0) {
                s_prefix = (prefix << bits[pass]) | (unsigned)bin;
                s_remaining = rem - cum;  // how many of the elements inside this bin are still to be taken
            }
        }
        __syncthreads();
    }
    const unsigned pivot = s_prefix;      // the top_n-th smallest key (32-bit score part)
    const int take_eq = (int)s_remaining;  // elements with key == pivot to keep, lowest indices first
    uint64_t* dst = sel_keys + (size_t)b * n_sel_pad;
    // wave w owns indices [w * per, (w + 1) * per), per a multiple of 64
    const int per = ((n + 15) / 16 + 63) / 64 * 64;
    const int lo = wave * per, hi = lo + per < n ? lo + per : n;
    int c_sel = 0, c_eq = 0;   // wave-uniform
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        unsigned key = 0xffffffffu;
        if (i < hi) key = ~ord_f32(sc[i]);
        c_sel += __popcll(__ballot(i < hi && key < pivot));
        c_eq += __popcll(__ballot(i < hi && key == pivot));
    }
    if (lane == 0) {
        s_wsel[wave] = c_sel;
        s_weq[wave] = c_eq;
    }
    __syncthreads();
    int base_eq = 0, base_less = 0;
    for (int w = 0; w < wave; ++w) {
        base_eq += s_weq[w];
        base_less += s_wsel[w];
    }
    // output position of an element = (selected elements before it): less-than-pivot ones before it + tied ones before it that are taken
    int run_less = base_less, run_eq = base_eq;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        unsigned key = 0xffffffffu;
        bool eq = false, less = false;
        if (i < hi) {
            key = ~ord_f32(sc[i]);
            eq = key == pivot;
            less = key < pivot;
        }
        const unsigned long long meq = __ballot(eq), mless = __ballot(less);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int eq_rank = run_eq + __popcll(meq & below);
        const bool sel = less || (eq && eq_rank < take_eq);
        const int eq_taken_before = eq_rank < take_eq ? eq_rank : take_eq;   // tied elements before this one that were taken
        const int pos = run_less + __popcll(mless & below) + eq_taken_before;
        if (sel) dst[pos] = ((uint64_t)key << 32) | (uint32_t)i;
        run_less += __popcll(mless);
        run_eq += __popcll(meq);
    }
    for (int i = top_n + tid; i < n_sel_pad; i += 1024) dst[i] = ~0ull;
}

// keys[b][i] = (~ord(score) << 32) | i for i < n (descending score, ascending index), ~0 for padding
__global__ void make_keys_kernel(const float* __restrict__ scores, int n, int n_pad, uint64_t* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    const int b = blockIdx.y;
    uint64_t k = ~0ull;
    if (i < n) k = ((uint64_t)(uint32_t)~ord_f32(scores[(size_t)b * n + i]) << 32) | (uint32_t)i;
    keys[(size_t)b * n_pad + i] = k;
}

// order[b][r] = index part of the r-th key; sorted_scores[b][r] = scores[b][order] (r < n_out)
__global__ void unpack_keys_kernel(const uint64_t* __restrict__ keys, int n_pad, const float* __restrict__ scores, int n,
                                   int n_out, int* __restrict__ order, float* __restrict__ sorted_scores) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_out) return;
    const int b = blockIdx.y;
    const uint64_t k = keys[(size_t)b * n_pad + r];
    const int idx = (int)(uint32_t)k;
    order[(size_t)b * n_out + r] = idx;
    if (sorted_scores) sorted_scores[(size_t)b * n_out + r] = scores[(size_t)b * n + idx];
}

// ---------------------------------------------------------------------------------------------- NMS
enum NmsMode : int { NMS_RPN = 0, NMS_HARD0 = 1, NMS_SOFT_LINEAR = 2, NMS_SOFT_GAUSS = 3, NMS_RETINA = 4 };

struct NmsArgs {
    const float* box_base;  // box of original index i = box_base + b*box_batch_stride + i*box_stride (x1,y1,x2,y2)
    long box_batch_stride;
    int box_stride;
    const float* classes;   // [B][n_src] or nullptr
    int n_src;
    const int* order;       // [B][n_cap]: sorted rank -> original index
    float* scores;          // [B][n_cap]: sorted scores, updated in place
    const int* n_dev;       // optional per-image element count on the device
    int n_cap;
    float thresh;
    int mode;
    int stop_after;         // hard modes: stop once this many boxes are kept (only the first n_out kept boxes are emitted,
                            // in score order, so what happens to later ones cannot change the result); 0 = run to the end
};

__device__ __forceinline__ float4 ldbox(const float* p) {
    return make_float4(p[0], p[1], p[2], p[3]);
}

// RpnNms.cu:38-48 / BatchedNms.cu:42-52 (i = later box, m = suppressor)
__device__ __forceinline__ float iou_plain(const float4 ib, const float4 mb) {
    const float x1 = ib.x > mb.x ? ib.x : mb.x;
    const float y1 = ib.y > mb.y ? ib.y : mb.y;
    const float x2 = ib.z < mb.z ? ib.z : mb.z;
    const float y2 = ib.w < mb.w ? ib.w : mb.w;
    float w = x2 - x1;
    w = w > 0.0f ? w : 0.0f;
    float h = y2 - y1;
    h = h > 0.0f ? h : 0.0f;
    const float iarea = (ib.z - ib.x) * (ib.w - ib.y);
    const float marea = (mb.z - mb.x) * (mb.w - mb.y);
    const float inter = w * h;
    return inter / (iarea + marea - inter);
}

// retinaface/common.hpp:91-104 (l = kept item, r = later box)
__device__ __forceinline__ float iou_retina(const float4 l, const float4 r) {
    const float i0 = l.x > r.x ? l.x : r.x;
    const float i1 = l.z < r.z ? l.z : r.z;
    const float i2 = l.y > r.y ? l.y : r.y;
    const float i3 = l.w < r.w ? l.w : r.w;
    if (i2 > i3 || i0 > i1) return 0.0f;
    const float inter = (i1 - i0) * (i3 - i2);
    return inter / ((l.z - l.x) * (l.w - l.y) + (r.z - r.x) * (r.w - r.y) - inter + 0.000001f);
}

__device__ __forceinline__ bool is_active(float s, int mode) {
    return (mode == NMS_RPN || mode == NMS_RETINA) ? (s > -FLT_MAX) : (s > 0.0f);
}

// effect of suppressor m (box mb, class mc) on element (box ib, class ic, score s)
__device__ __forceinline__ float apply_one(float s, const float4 ib, int ic, const float4 mb, int mc, float thresh, int mode) {
    if (mode == NMS_RETINA) {
        if (iou_retina(mb, ib) > thresh) s = -FLT_MAX;
        return s;
    }
    if (mode != NMS_RPN && ic != mc) return s;
    const float ov = iou_plain(ib, mb);
    if (!(ov > thresh)) return s;
    switch (mode) {
        case NMS_RPN: return -FLT_MAX;
        case NMS_SOFT_LINEAR: return (1 - ov) * s;
        case NMS_SOFT_GAUSS: {
            const float sigma = 0.5;
            return expf(-(ov * ov) / sigma) * s;
        }
        default: return 0.0f;
    }
}

__global__ __launch_bounds__(1024) void greedy_nms_kernel(NmsArgs a) {
    __shared__ float4 s_box[64];
    __shared__ int s_cls[64];
    __shared__ int s_act[64];
    __shared__ int s_kept;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_kept = 0;
    int n = a.n_cap;
    if (a.n_dev) {
        const int nd = a.n_dev[b];
        n = nd < n ? nd : n;
    }
    const float* boxes = a.box_base + (size_t)b * a.box_batch_stride;
    const float* cls = a.classes ? a.classes + (size_t)b * a.n_src : nullptr;
    const int* order = a.order + (size_t)b * a.n_cap;
    float* scores = a.scores + (size_t)b * a.n_cap;
    const bool hard = a.mode == NMS_RPN || a.mode == NMS_RETINA || a.mode == NMS_HARD0;
    const int nblk = (n + 63) >> 6;
    for (int bi = 0; bi < nblk; ++bi) {
        const int base = bi << 6;
        if (wave == 0) {
            const int r = base + lane;
            const bool have = r < n;
            float4 mb = make_float4(0.f, 0.f, 0.f, 0.f);
            int mc = 0;
            float ms = -FLT_MAX;
            if (have) {
                const int idx = order[r];
                mb = ldbox(boxes + (size_t)idx * a.box_stride);
                mc = cls ? (int)cls[idx] : 0;
                ms = scores[r];
            }
            for (int k = 0; k < 63; ++k) {
                const float sk = __shfl(ms, k);
                if (base + k >= n) break;                // wave-uniform
                if (!is_active(sk, a.mode)) continue;    // wave-uniform
                const float4 kb = make_float4(__shfl(mb.x, k), __shfl(mb.y, k), __shfl(mb.z, k), __shfl(mb.w, k));
                const int kc = __shfl(mc, k);
                if (have && lane > k && !(hard && !is_active(ms, a.mode))) ms = apply_one(ms, mb, mc, kb, kc, a.thresh, a.mode);
            }
            if (have) scores[r] = ms;
            s_box[lane] = mb;
            s_cls[lane] = mc;
            const bool act = have && is_active(ms, a.mode);
            s_act[lane] = act ? 1 : 0;
            const unsigned long long am = __ballot(act);
            if (lane == 0) s_kept += __popcll(am);
        }
        __syncthreads();
        if (hard && a.stop_after > 0 && s_kept >= a.stop_after) break;  // uniform: every thread reads the same count
        for (int r = base + 64 + tid; r < n; r += 1024) {
            float s = scores[r];
            if (hard && !is_active(s, a.mode)) continue;
            const int idx = order[r];
            const float4 ib = ldbox(boxes + (size_t)idx * a.box_stride);
            const int ic = cls ? (int)cls[idx] : 0;
            for (int k = 0; k < 64; ++k) {
                if (!s_act[k]) continue;
                s = apply_one(s, ib, ic, s_box[k], s_cls[k], a.thresh, a.mode);
                if (hard && !is_active(s, a.mode)) break;
            }
            scores[r] = s;
        }
        __syncthreads();
    }
}

// The same walk for n <= 1024 (BatchedNms: 1000 boxes per image, rcnn/BatchedNms.cu:28-88; soft-NMS cannot use the bit matrix below - a
// suppressed box stays a suppressor with a decayed score).  The element of sorted rank t lives in thread t's REGISTERS for the whole walk -
// box, class, score loaded once, the score stored once - where the kernel above goes to global memory twice per 64-box block (order -> box
// -> score, then the scores back): 16 blocks x ~6 dependent round trips were most of its 672 us at C5 (profiles/r04_kernel_stats_c5_1ctx_lanes1.txt).
// Wave w resolves block w among its own lanes by shuffles, publishes the block, every later rank applies it from LDS: the same
// suppressors in the same order as above, hence the same bits.
__global__ __launch_bounds__(1024) void greedy_nms_small_kernel(NmsArgs a) {
    __shared__ float4 s_box[64];
    __shared__ int s_cls[64];
    __shared__ int s_act[64];
    __shared__ int s_kept;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_kept = 0;
    int n = a.n_cap;
    if (a.n_dev) {
        const int nd = a.n_dev[b];
        n = nd < n ? nd : n;
    }
    const float* boxes = a.box_base + (size_t)b * a.box_batch_stride;
    const float* cls = a.classes ? a.classes + (size_t)b * a.n_src : nullptr;
    const int* order = a.order + (size_t)b * a.n_cap;
    float* scores = a.scores + (size_t)b * a.n_cap;
    const bool hard = a.mode == NMS_RPN || a.mode == NMS_RETINA || a.mode == NMS_HARD0;
    const int nblk = (n + 63) >> 6;
    const int r = tid;
    const bool have = r < n;
    float4 mb = make_float4(0.f, 0.f, 0.f, 0.f);
    int mc = 0;
    float ms = -FLT_MAX;
    if (have) {
        const int idx = order[r];
        mb = ldbox(boxes + (size_t)idx * a.box_stride);
        mc = cls ? (int)cls[idx] : 0;
        ms = scores[r];
    }
    __syncthreads();
    for (int bi = 0; bi < nblk; ++bi) {
        const int base = bi << 6;
        if (wave == bi) {
            for (int k = 0; k < 63; ++k) {
                const float sk = __shfl(ms, k);
                if (base + k >= n) break;                // wave-uniform
                if (!is_active(sk, a.mode)) continue;    // wave-uniform
                const float4 kb = make_float4(__shfl(mb.x, k), __shfl(mb.y, k), __shfl(mb.z, k), __shfl(mb.w, k));
                const int kc = __shfl(mc, k);
                if (have && lane > k && !(hard && !is_active(ms, a.mode))) ms = apply_one(ms, mb, mc, kb, kc, a.thresh, a.mode);
            }
            s_box[lane] = mb;
            s_cls[lane] = mc;
            const bool act = have && is_active(ms, a.mode);
            s_act[lane] = act ? 1 : 0;
            const unsigned long long am = __ballot(act);
            if (lane == 0) s_kept += __popcll(am);
        }
        __syncthreads();
        if (hard && a.stop_after > 0 && s_kept >= a.stop_after) break;  // uniform: every thread reads the same count
        if (have && r >= base + 64 && !(hard && !is_active(ms, a.mode))) {
            for (int k = 0; k < 64; ++k) {
                if (!s_act[k]) continue;
                ms = apply_one(ms, mb, mc, s_box[k], s_cls[k], a.thresh, a.mode);
                if (hard && !is_active(ms, a.mode)) break;
            }
        }
        __syncthreads();
    }
    if (have) scores[r] = ms;
}

// ---- BatchedNms (hard / soft-linear / soft-gaussian, class-aware; rcnn/BatchedNms.cu:28-88) with the pair work done up front -------------
// Measured (round 4): moving the walk's state from global memory into registers (kernel above) left its 690 us at C5 unchanged - the walk
// is 1000 DEPENDENT steps, and each step's chain (IoU with a correctly rounded division, the soft weight, ~80 instructions a lone wave
// issues back to back) is what costs.  But only the SCORE update is sequential: whether suppressor k touches box i (same class, IoU above the
// threshold) and by what factor - (1 - iou), exp(-iou^2 / sigma), or "to zero" - depends on the two boxes alone.  soft_factor_kernel
// computes that factor for every (suppressor, later box) pair with the whole chip, one wave per 64 x 64 tile, into f_t[k][i] (k-major:
// the walk reads a row of 64 consecutive boxes per load; -1 = no effect); the walk then is `s = f * s`, one multiply per step, the loads
// independent of the scores.  The same operands in the same order as apply_one(): the same bits.
__global__ __launch_bounds__(64) void soft_factor_kernel(const float* __restrict__ box_base, long box_batch_stride, const float* __restrict__ classes,
                                                         int n_src, const int* __restrict__ order, int n, int n_blk, float thresh, int mode,
                                                         float* __restrict__ f_t) {
    const int c = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
    if (c > r) return;
    const int lane = threadIdx.x;
    const float* boxes = box_base + (size_t)b * box_batch_stride;
    const float* cls = classes + (size_t)b * n_src;
    const int* ord = order + (size_t)b * n;
    __shared__ float4 s_box[64];
    __shared__ int s_cls[64];
    const int j0 = c * 64 + lane;
    s_box[lane] = j0 < n ? ldbox(boxes + (size_t)ord[j0] * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s_cls[lane] = j0 < n ? (int)cls[ord[j0]] : -1;
    __syncthreads();
    const int i = r * 64 + lane;
    const bool have = i < n;
    const float4 ib = have ? ldbox(boxes + (size_t)ord[i] * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int ic = have ? (int)cls[ord[i]] : -2;
    const size_t row_len = (size_t)n_blk * 64;
    float* dst = f_t + ((size_t)b * n_blk + c) * 64 * row_len + i;
    const int kend = (c == r) ? 63 : 64;   // (the diagonal tile's last suppressor has no later box in its block)
    for (int k = 0; k < kend; ++k) {
        float f = -1.0f;
        if (have && (c != r || lane > k) && ic == s_cls[k]) {
            const float ov = iou_plain(ib, s_box[k]);
            if (ov > thresh) {
                if (mode == NMS_SOFT_LINEAR) f = 1 - ov;
                else if (mode == NMS_SOFT_GAUSS) {
                    const float sigma = 0.5;
                    f = expf(-(ov * ov) / sigma);
                } else f = 0.0f;
            }
        }
        dst[(size_t)k * row_len] = f;
    }
}

// the walk of greedy_nms_small_kernel over precomputed factors (modes NMS_HARD0 / NMS_SOFT_LINEAR / NMS_SOFT_GAUSS, n <= 1024)
__global__ __launch_bounds__(1024) void greedy_nms_factor_kernel(float* __restrict__ scores_all, int n, int n_blk, const float* __restrict__ f_t,
                                                                 int mode, int stop_after) {
    __shared__ int s_act[64];
    __shared__ int s_kept;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_kept = 0;
    float* scores = scores_all + (size_t)b * n;
    const bool hard = mode == NMS_HARD0;
    const int r = tid;
    const bool have = r < n;
    float ms = have ? scores[r] : -FLT_MAX;
    const size_t row_len = (size_t)n_blk * 64;
    const float* fb = f_t + (size_t)b * n_blk * 64 * row_len + r;
    __syncthreads();
    for (int bi = 0; bi < n_blk; ++bi) {
        const int base = bi << 6;
        const float* frow = fb + (size_t)base * row_len;   // factor of suppressor base + k on this thread's box: frow[k * row_len]
        if (wave == bi) {
            float fk[63];
#pragma unroll
            // unconditional: soft_factor_kernel wrote -1 for the lanes a suppressor does not reach (lane <= k, boxes past n).  Written as
            // `cond ? load : -1` every one of the 63 loads sits in its own exec-masked branch with a full wait behind it - 63 L2 round
            // trips in series per block, 600 us per launch (measured; the same trap as the stem kernel's weight loads)
            for (int k = 0; k < 63; ++k) fk[k] = frow[(size_t)k * row_len];
#pragma unroll
            for (int k = 0; k < 63; ++k) {
                const float sk = __shfl(ms, k);
                if (base + k >= n) break;                // wave-uniform
                if (!(sk > 0.0f)) continue;              // wave-uniform: an inactive box suppresses nothing
                if (fk[k] >= 0.0f && !(hard && !(ms > 0.0f))) ms = hard ? 0.0f : fk[k] * ms;
            }
            const bool act = have && ms > 0.0f;
            s_act[lane] = act ? 1 : 0;
            const unsigned long long am = __ballot(act);
            if (lane == 0) s_kept += __popcll(am);
        }
        __syncthreads();
        if (hard && stop_after > 0 && s_kept >= stop_after) break;  // uniform: every thread reads the same count
        if (have && r >= base + 64 && !(hard && !(ms > 0.0f))) {
            float fk[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) fk[k] = frow[(size_t)k * row_len];
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                if (!s_act[k]) continue;
                if (fk[k] >= 0.0f) ms = hard ? 0.0f : fk[k] * ms;
                if (hard && !(ms > 0.0f)) break;
            }
        }
        __syncthreads();
    }
    if (have) scores[r] = ms;
}

// ---- hard class-agnostic NMS as suppression bit-matrix + scan (RPN: 6000 sorted boxes, first 1000 kept) ---------------
// The greedy kernel above walks 94 blocks of 64 boxes and, per block, makes every later box test up to 64 IoUs:
// 3.6-5.3 ms per image on the R-CNN config.  Here the IoU work is done once, by the whole chip:
//   hard_mask_kernel  one wave per 64x64 tile (column block <= row block): bit k of mask_t[c][i] says "sorted box 64c+k
//                     (earlier, higher score) overlaps sorted box i by more than the threshold".  Column-block-major
//                     layout so that the scan reads one contiguous run of words per step.
//   hard_scan_kernel  one workgroup per image; thread t owns sorted ranks t, t+1024, ... and keeps their "removed" flags in
//                     a register.  Step c: the wave owning block c resolves it (in-block chain over live suppressors
//                     only), publishes the keep word, then every thread ANDs word c of its later rows against it.
//                     Stops as soon as `stop_after` boxes are kept.  Writes -FLT_MAX into the scores of removed boxes,
//                     exactly what greedy_nms_kernel leaves behind, so the re-sort / gather tail is shared.
constexpr int kMaskMaxN = 8192;

__global__ __launch_bounds__(64) void hard_mask_kernel(const float* __restrict__ box_base, long box_batch_stride,
                                                       const int* __restrict__ order, int n, int n_blk, float thresh,
                                                       uint64_t* __restrict__ mask_t) {
    const int c = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
    if (c > r) return;
    const int lane = threadIdx.x;
    const float* boxes = box_base + (size_t)b * box_batch_stride;
    const int* ord = order + (size_t)b * n;
    __shared__ float4 s_box[64];
    const int j0 = c * 64 + lane;
    s_box[lane] = j0 < n ? ldbox(boxes + (size_t)ord[j0] * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int i = r * 64 + lane;
    uint64_t bits = 0;
    if (i < n) {
        const float4 ib = ldbox(boxes + (size_t)ord[i] * 4);
        const int kend = (c == r) ? lane : 64;  // only earlier boxes suppress
        for (int k = 0; k < kend; ++k)
            if (iou_plain(ib, s_box[k]) > thresh) bits |= 1ull << k;  // same call as apply_one(NMS_RPN): (later, suppressor)
    }
    mask_t[((size_t)b * n_blk + c) * ((size_t)n_blk * 64) + i] = bits;
}

__global__ __launch_bounds__(1024) void hard_scan_kernel(const uint64_t* __restrict__ mask_t, float* __restrict__ scores, int n,
                                                         int n_blk, int stop_after) {
    constexpr int kSlots = kMaskMaxN / 1024;
    __shared__ uint64_t s_keep[kMaskMaxN / 64];
    // kept count AFTER block c, one slot per block: written once by the wave that owns block c before the barrier of iteration c
    // and never again, so every thread's stop decision of iteration c reads the same value no matter how far other waves have run
    // ahead (a single running counter could already hold iteration c+1's update when a slow wave gets to read it)
    __shared__ int s_cum[kMaskMaxN / 64];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sc = scores + (size_t)b * n;
    const uint64_t* mt = mask_t + (size_t)b * n_blk * ((size_t)n_blk * 64);
    const size_t row_len = (size_t)n_blk * 64;
    unsigned rem = 0;  // bit s: rank tid + 1024*s is removed (or padding / inactive from the start)
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int i = tid + 1024 * s;
        if (i >= n || !(sc[i] > -FLT_MAX)) rem |= 1u << s;
    }
    __syncthreads();
    int processed = 0;
    for (int c = 0; c < n_blk; ++c) {
        const int slot = c >> 4;  // block c = ranks 64c .. 64c+63 = threads of wave c % 16, slot c / 16
        if (wave == (c & 15)) {
            const uint64_t supby = mt[(size_t)c * row_len + c * 64 + lane];  // diagonal tile: suppressors inside the block
            uint64_t colany = supby;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const uint32_t lo = __shfl_xor((int)(uint32_t)colany, m), hi = __shfl_xor((int)(uint32_t)(colany >> 32), m);
                colany |= ((uint64_t)hi << 32) | lo;
            }
            uint64_t dead = __ballot((rem >> slot) & 1u);
            uint64_t todo = colany & ~dead;
            while (todo) {  // ascending over live boxes that suppress something: the sequential greedy order
                const int k = __ffsll((unsigned long long)todo) - 1;
                const uint64_t col = __ballot((supby >> k) & 1ull);
                dead |= col;
                todo &= ~col & ~(1ull << k);
            }
            if ((dead >> lane) & 1ull) rem |= 1u << slot;
            if (lane == 0) {
                s_keep[c] = ~dead;
                s_cum[c] = (c ? s_cum[c - 1] : 0) + __popcll(~dead);
            }
        }
        __syncthreads();
        processed = c + 1;
        if (stop_after > 0 && s_cum[c] >= stop_after) break;
        const uint64_t keep = s_keep[c];
        if (keep) {
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                const int i = tid + 1024 * s;
                if (i >= (c + 1) * 64 && i < n && !((rem >> s) & 1u) && (mt[(size_t)c * row_len + i] & keep)) rem |= 1u << s;
            }
        }
    }
    // leave the scores the way the sequential pass would: removed boxes of the processed prefix get -FLT_MAX
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int i = tid + 1024 * s;
        if (i < n && i < processed * 64 && ((rem >> s) & 1u)) sc[i] = -FLT_MAX;
    }
}

// ---------------------------------------------------------------------------------------------- RetinaFace
constexpr int kRfDet = 15;

struct RfLevels {
    const float* in[3];
    int cell_off[4];
    int w[3], h[3], anchor[3];
};

// decode.cu:110-165 for one (cell, k); returns whether the anchor passes conf2 > 0.02 (and its conf2)
__device__ __forceinline__ bool rf_conf(const float* cls_reg, int idx, int k, int total, float* conf_out) {
    const float conf1 = cls_reg[idx + k * total * 2];
    float conf2 = cls_reg[idx + k * total * 2 + total];
    conf2 = expf(conf2) / (expf(conf1) + expf(conf2));
    *conf_out = conf2;
    return !((double)conf2 <= 0.02);
}

constexpr int kRfChunk = 512;  // cells per workgroup

__global__ __launch_bounds__(kRfChunk) void retina_count_kernel(RfLevels t, int total_cells, int* __restrict__ chunk_cnt,
                                                                int n_chunks) {
    const int b = blockIdx.y;
    const int g = blockIdx.x * kRfChunk + threadIdx.x;
    int c = 0;
    if (g < total_cells) {
        const int l = g >= t.cell_off[2] ? 2 : (g >= t.cell_off[1] ? 1 : 0);
        const int total = t.w[l] * t.h[l];
        const int idx = g - t.cell_off[l];
        const float* cls_reg = t.in[l] + (size_t)b * 32 * total + 2 * 4 * total;
        float cf;
        c += rf_conf(cls_reg, idx, 0, total, &cf) ? 1 : 0;
        c += rf_conf(cls_reg, idx, 1, total, &cf) ? 1 : 0;
    }
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int w = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&s_cnt, w);
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[b * n_chunks + blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(kRfChunk) void retina_emit_kernel(RfLevels t, int total_cells, const int* __restrict__ chunk_cnt,
                                                               int n_chunks, int net_h, int net_w, int out_elem,
                                                               float* __restrict__ output) {
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = kRfChunk / 64;
    __shared__ int s_wave[kWaves];
    __shared__ int s_base;
    if (wave == 0) {
        int acc = 0;
        for (int j = lane; j < chunk; j += 64) acc += chunk_cnt[b * n_chunks + j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) s_base = acc;
    }
    const int g = chunk * kRfChunk + threadIdx.x;
    bool f0 = false, f1 = false;
    float c0 = 0.f, c1 = 0.f;
    int l = 0, total = 1, idx = 0;
    const float* cur = nullptr;
    if (g < total_cells) {
        l = g >= t.cell_off[2] ? 2 : (g >= t.cell_off[1] ? 1 : 0);
        total = t.w[l] * t.h[l];
        idx = g - t.cell_off[l];
        cur = t.in[l] + (size_t)b * 32 * total;
        const float* cls_reg = cur + 2 * 4 * total;
        f0 = rf_conf(cls_reg, idx, 0, total, &c0);
        f1 = rf_conf(cls_reg, idx, 1, total, &c1);
    }
    // exclusive scan of (f0 + f1) over the workgroup
    const int mine = (f0 ? 1 : 0) + (f1 ? 1 : 0);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = s_base;
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv)
        if (wv < wave) before += s_wave[wv];
    int slot = before + incl - mine;
    float* out = output + (size_t)b * out_elem;
    if (g < total_cells) {
        const int w = t.w[l], h = t.h[l];
        const int y = idx / w, x = idx - y * w;
        const float* bbox_reg = cur;
        const float* lmk_reg = cur + 2 * 4 * total + 2 * 2 * total;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!(k == 0 ? f0 : f1)) continue;
            float* det = out + 1 + (size_t)slot * kRfDet;
            ++slot;
            float prior[4];
            prior[0] = (float)(((double)(float)x + 0.5) / w);
            prior[1] = (float)(((double)(float)y + 0.5) / h);
            prior[2] = (float)t.anchor[l] * (k + 1) / net_w;
            prior[3] = (float)t.anchor[l] * (k + 1) / net_h;
            float bb0 = (float)((double)prior[0] + (double)bbox_reg[idx + k * total * 4] * 0.1 * (double)prior[2]);
            float bb1 = (float)((double)prior[1] + (double)bbox_reg[idx + k * total * 4 + total] * 0.1 * (double)prior[3]);
            float bb2 = prior[2] * expf((float)((double)bbox_reg[idx + k * total * 4 + total * 2] * 0.2));
            float bb3 = prior[3] * expf((float)((double)bbox_reg[idx + k * total * 4 + total * 3] * 0.2));
            bb0 -= bb2 / 2;
            bb1 -= bb3 / 2;
            bb2 += bb0;
            bb3 += bb1;
            bb0 *= net_w;
            bb1 *= net_h;
            bb2 *= net_w;
            bb3 *= net_h;
            det[0] = bb0;
            det[1] = bb1;
            det[2] = bb2;
            det[3] = bb3;
            det[4] = k == 0 ? c0 : c1;
#pragma unroll
            for (int i = 0; i < 10; i += 2) {
                float lx = (float)((double)prior[0] + (double)lmk_reg[idx + k * total * 10 + total * i] * 0.1 * (double)prior[2]);
                float ly = (float)((double)prior[1] + (double)lmk_reg[idx + k * total * 10 + total * (i + 1)] * 0.1 * (double)prior[3]);
                lx *= net_w;
                ly *= net_h;
                det[5 + i] = lx;
                det[5 + i + 1] = ly;
            }
        }
    }
    if (chunk == n_chunks - 1 && threadIdx.x == kRfChunk - 1) out[0] = (float)(before + incl);
}

// keys over the decode buffer: slot i < count with conf > thresh (double compare, common.hpp:113)
__global__ void retina_keys_kernel(const float* __restrict__ dec, int out_elem, int n_pad, double conf_thresh,
                                   uint64_t* __restrict__ keys, int* __restrict__ n_valid) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* img = dec + (size_t)b * out_elem;
    const int count = (int)img[0];
    uint64_t k = ~0ull;
    bool ok = false;
    if (i < count && i < n_pad) {
        const float c = img[1 + (size_t)i * kRfDet + 4];
        ok = !((double)c <= conf_thresh);
        if (ok) k = ((uint64_t)(uint32_t)~ord_f32(c) << 32) | (uint32_t)i;
    }
    if (i < n_pad) keys[(size_t)b * n_pad + i] = k;
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&n_valid[b], __popcll(m));
}

__global__ void retina_unpack_kernel(const uint64_t* __restrict__ keys, int n_pad, const float* __restrict__ dec,
                                     int out_elem, int* __restrict__ order, float* __restrict__ scores) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    const uint64_t k = keys[(size_t)b * n_pad + r];
    const bool valid = k != ~0ull;
    const int idx = valid ? (int)(uint32_t)k : 0;
    order[(size_t)b * n_pad + r] = idx;
    scores[(size_t)b * n_pad + r] = valid ? dec[(size_t)b * out_elem + 1 + (size_t)idx * kRfDet + 4] : -FLT_MAX;
}

// ordered compaction of the survivors (score > -FLT_MAX) of the first n sorted elements
__global__ __launch_bounds__(1024) void compact_kept_kernel(const float* __restrict__ scores, const int* __restrict__ order,
                                                            const int* __restrict__ n_dev, int n_cap, int max_keep,
                                                            int* __restrict__ keep_idx, int* __restrict__ keep_cnt,
                                                            const float* __restrict__ rec_base, long rec_batch_stride,
                                                            int rec_floats, float* __restrict__ keep_rec) {
    __shared__ int s_wave[16];
    __shared__ int s_run;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n = n_dev ? n_dev[b] : n_cap;
    n = n < n_cap ? n : n_cap;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int r = base + tid;
        const bool keep = r < n && scores[(size_t)b * n_cap + r] > -FLT_MAX;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int pos = s_run + __popcll(m & ((1ull << lane) - 1ull));
        int tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) pos += s_wave[w];
            tot += s_wave[w];
        }
        if (keep && pos < max_keep) {
            const int idx = order[(size_t)b * n_cap + r];
            keep_idx[(size_t)b * max_keep + pos] = idx;
            if (keep_rec)
                for (int e = 0; e < rec_floats; ++e)
                    keep_rec[((size_t)b * max_keep + pos) * rec_floats + e] = rec_base[(size_t)b * rec_batch_stride + (size_t)idx * rec_floats + e];
        }
        __syncthreads();
        if (tid == 0) s_run += tot;
        __syncthreads();
    }
    if (tid == 0) keep_cnt[b] = s_run < max_keep ? s_run : max_keep;
}

// ---------------------------------------------------------------------------------------------- R-CNN kernels
// RpnDecode.cu:90-133
__global__ void rpn_decode_kernel(const int* __restrict__ order, int n_order, int num, int top_n, const float* __restrict__ scores,
                                  const float* __restrict__ deltas, int scores_size, int height, int width, int num_anchors,
                                  float stride, const float* __restrict__ anchors, float image_w, float image_h,
                                  float* __restrict__ out_scores, float* __restrict__ out_boxes) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (d >= top_n) return;
    float* os = out_scores + (size_t)b * top_n;
    float* ob = out_boxes + ((size_t)b * top_n + d) * 4;
    if (d >= num) {
        os[d] = -FLT_MAX;
        ob[0] = ob[1] = ob[2] = ob[3] = 0.0f;
        return;
    }
    const int i = order ? order[(size_t)b * n_order + d] : d;
    const float* in_scores = scores + (size_t)b * scores_size;
    const float* in_boxes = deltas + (size_t)b * scores_size * 4;
    const int x = i % width;
    const int y = (i / width) % height;
    const int a = (i / height / width) % num_anchors;
    const float bx = in_boxes[((a * 4 + 0) * height + y) * width + x];
    const float by = in_boxes[((a * 4 + 1) * height + y) * width + x];
    const float bz = in_boxes[((a * 4 + 2) * height + y) * width + x];
    const float bw = in_boxes[((a * 4 + 3) * height + y) * width + x];
    const float fx = x * stride, fy = y * stride;
    const float* dd = anchors + 4 * a;
    const float x1 = fx + dd[0], y1 = fy + dd[1], x2 = fx + dd[2], y2 = fy + dd[3];
    const float w = x2 - x1, h = y2 - y1;
    const float pcx = bx * w + x1 + 0.5f * w;
    const float pcy = by * h + y1 + 0.5f * h;
    const float pw = expf(bz) * w;
    const float ph = expf(bw) * h;
    float r0 = pcx - 0.5f * pw;
    r0 = r0 > 0.0f ? r0 : 0.0f;
    float r1 = pcy - 0.5f * ph;
    r1 = r1 > 0.0f ? r1 : 0.0f;
    float r2 = pcx + 0.5f * pw;
    r2 = r2 < image_w ? r2 : image_w;
    float r3 = pcy + 0.5f * ph;
    r3 = r3 < image_h ? r3 : image_h;
    ob[0] = r0;
    ob[1] = r1;
    ob[2] = r2;
    ob[3] = r3;
    os[d] = (r2 - r0 <= 0.0f || r3 - r1 <= 0.0f) ? -FLT_MAX : in_scores[i];
}

// keys for the re-sort after NMS: (~ord(updated score) << 32) | sorted position
__global__ void rekey_kernel(const float* __restrict__ scores, int n, int n_pad, uint64_t* __restrict__ keys) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    const int b = blockIdx.y;
    uint64_t k = ~0ull;
    if (r < n) k = ((uint64_t)(uint32_t)~ord_f32(scores[(size_t)b * n + r]) << 32) | (uint32_t)r;
    keys[(size_t)b * n_pad + r] = k;
}

// out[d] = record of original index order1[pos2[d]] (RpnNms.cu:116, BatchedNms.cu:150-158)
__global__ void gather_after_nms_kernel(const uint64_t* __restrict__ keys2, int n_pad, const int* __restrict__ order1, int n,
                                        const float* __restrict__ scores_sorted, const float* __restrict__ boxes,
                                        const float* __restrict__ classes, int n_out, float* __restrict__ out_scores,
                                        float* __restrict__ out_boxes, float* __restrict__ out_classes) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (d >= n_out) return;
    float sc = 0.0f, cl = 0.0f;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d < n) {
        const int pos = (int)(uint32_t)keys2[(size_t)b * n_pad + d];
        const int idx = order1[(size_t)b * n + pos];
        sc = scores_sorted[(size_t)b * n + pos];
        bx = ldbox(boxes + ((size_t)b * n + idx) * 4);
        if (classes) cl = classes[(size_t)b * n + idx];
    }
    if (out_scores) out_scores[(size_t)b * n_out + d] = sc;
    float* ob = out_boxes + ((size_t)b * n_out + d) * 4;
    ob[0] = bx.x;
    ob[1] = bx.y;
    ob[2] = bx.z;
    ob[3] = bx.w;
    if (out_classes) out_classes[(size_t)b * n_out + d] = cl;
}

// PredictorDecode.cu:79-106
__global__ void predictor_decode_kernel(const int* __restrict__ order, int n_order, const float* __restrict__ scores,
                                        const float* __restrict__ deltas, const float* __restrict__ proposals, int num_boxes,
                                        int num_classes, float image_w, float w0, float w1, float w2, float w3,
                                        float* __restrict__ out_scores, float* __restrict__ out_boxes,
                                        float* __restrict__ out_classes) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (d >= num_boxes) return;
    const int scores_size = num_boxes * num_classes;
    const int i = order[(size_t)b * n_order + d];
    const int cls = i % num_classes, n = i / num_classes;
    const float* dl = deltas + ((size_t)b * scores_size + i) * 4;
    const float* bx = proposals + ((size_t)b * num_boxes + n) * 4;
    const float w = bx[2] - bx[0], h = bx[3] - bx[1];
    const float pcx = (dl[0] / w0) * w + bx[0] + 0.5f * w;
    const float pcy = (dl[1] / w1) * h + bx[1] + 0.5f * h;
    const float pw = expf(dl[2] / w2) * w;
    const float ph = expf(dl[3] / w3) * h;
    float r0 = pcx - 0.5f * pw;
    r0 = r0 > 0.0f ? r0 : 0.0f;
    float r1 = pcy - 0.5f * ph;
    r1 = r1 > 0.0f ? r1 : 0.0f;
    float r2 = pcx + 0.5f * pw;
    r2 = r2 < image_w ? r2 : image_w;
    float r3 = pcy + 0.5f * ph;
    r3 = r3 < image_w ? r3 : image_w;  // reference clips y2 with image_width (PredictorDecode.cu:99): kept for parity
    float* ob = out_boxes + ((size_t)b * num_boxes + d) * 4;
    ob[0] = r0;
    ob[1] = r1;
    ob[2] = r2;
    ob[3] = r3;
    out_scores[(size_t)b * num_boxes + d] = (r2 - r0 <= 0.0f || r3 - r1 <= 0.0f) ? 0.0f : scores[(size_t)b * scores_size + i];
    out_classes[(size_t)b * num_boxes + d] = (float)cls;
}

struct Carver {
    char* p;
    size_t left;
    bool ok = true;
    template <typename T>
    T* take(size_t count) {
        const size_t bytes = align_up(count * sizeof(T), 256);
        if (bytes > left) {
            ok = false;
            return nullptr;
        }
        T* r = reinterpret_cast<T*>(p);
        p += bytes;
        left -= bytes;
        return r;
    }
};

inline dim3 grid1(int n, int batch, int threads = 256) {
    return dim3((n + threads - 1) / threads, batch);
}

int rf_total_anchors(int net_h, int net_w) {
    int a = 0;
    for (int s = 8; s <= 32; s *= 2) a += (net_h / s) * (net_w / s) * 2;
    return a;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" size_t trtx_retina_decode_output_floats(int net_h, int net_w) {
    return 1 + (size_t)rf_total_anchors(net_h, net_w) * kRfDet;
}

extern "C" size_t trtx_retina_decode_workspace(int batch, int net_h, int net_w) {
    const int cells = rf_total_anchors(net_h, net_w) / 2;
    return align_up((size_t)batch * ((cells + kRfChunk - 1) / kRfChunk) * sizeof(int), 256);
}

extern "C" int32_t trtx_retina_decode(const float* const* inputs, int batch, int net_h, int net_w, float* output,
                                      void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!inputs || !output || !workspace || batch < 1 || net_h < 32 || net_w < 32) return TRTX_ERR_INVALID;
    if (workspace_bytes < trtx_retina_decode_workspace(batch, net_h, net_w)) return TRTX_ERR_WORKSPACE;
    RfLevels t{};
    int off = 0, step = 8, anchor = 16;
    for (int l = 0; l < 3; ++l, step *= 2, anchor *= 4) {
        t.in[l] = inputs[l];
        t.h[l] = net_h / step;
        t.w[l] = net_w / step;
        t.anchor[l] = anchor;
        t.cell_off[l] = off;
        off += t.h[l] * t.w[l];
    }
    t.cell_off[3] = off;
    const int n_chunks = (off + kRfChunk - 1) / kRfChunk;
    int* chunk_cnt = static_cast<int*>(workspace);
    const int out_elem = (int)trtx_retina_decode_output_floats(net_h, net_w);
    hipLaunchKernelGGL(retina_count_kernel, dim3(n_chunks, batch), dim3(kRfChunk), 0, stream, t, off, chunk_cnt, n_chunks);
    hipLaunchKernelGGL(retina_emit_kernel, dim3(n_chunks, batch), dim3(kRfChunk), 0, stream, t, off, chunk_cnt, n_chunks, net_h,
                       net_w, out_elem, output);
    return trtx::check_launch("trtx_retina_decode");
}

extern "C" size_t trtx_retina_nms_workspace(int batch, int net_h, int net_w) {
    const size_t n_pad = next_pow2(rf_total_anchors(net_h, net_w));
    return align_up(batch * n_pad * 8, 256) + 2 * align_up(batch * n_pad * 4, 256) + align_up((size_t)batch * 4, 256);
}

extern "C" int32_t trtx_retina_nms(const float* decode_out, int batch, int net_h, int net_w, double conf_thresh,
                                   float nms_thresh, int max_keep, int32_t* keep_idx, int32_t* keep_cnt, float* keep_det,
                                   void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!decode_out || !keep_idx || !keep_cnt || !workspace || batch < 1 || max_keep < 1) return TRTX_ERR_INVALID;
    const int anchors = rf_total_anchors(net_h, net_w);
    const int n_pad = next_pow2(anchors);
    const int out_elem = 1 + anchors * kRfDet;
    Carver c{static_cast<char*>(workspace), workspace_bytes};
    uint64_t* keys = c.take<uint64_t>((size_t)batch * n_pad);
    int* order = c.take<int>((size_t)batch * n_pad);
    float* scores = c.take<float>((size_t)batch * n_pad);
    int* n_valid = c.take<int>(batch);
    if (!c.ok) return TRTX_ERR_WORKSPACE;
    TRTX_HIP_TRY(hipMemsetAsync(n_valid, 0, sizeof(int) * batch, stream));
    hipLaunchKernelGGL(retina_keys_kernel, grid1(n_pad, batch), dim3(256), 0, stream, decode_out, out_elem, n_pad, conf_thresh,
                       keys, n_valid);
    int32_t st = sort_keys(keys, batch, n_pad, stream);
    if (st != TRTX_OK) return st;
    hipLaunchKernelGGL(retina_unpack_kernel, grid1(n_pad, batch), dim3(256), 0, stream, keys, n_pad, decode_out, out_elem, order,
                       scores);
    NmsArgs a{};
    a.box_base = decode_out + 1;
    a.box_batch_stride = out_elem;
    a.box_stride = kRfDet;
    a.order = order;
    a.scores = scores;
    a.n_dev = n_valid;
    a.n_cap = n_pad;
    a.thresh = nms_thresh;
    a.mode = NMS_RETINA;
    a.stop_after = max_keep;
    if (a.n_cap <= 1024) hipLaunchKernelGGL(greedy_nms_small_kernel, dim3(batch), dim3(1024), 0, stream, a);
    else hipLaunchKernelGGL(greedy_nms_kernel, dim3(batch), dim3(1024), 0, stream, a);
    hipLaunchKernelGGL(compact_kept_kernel, dim3(batch), dim3(1024), 0, stream, scores, order, n_valid, n_pad, max_keep, keep_idx,
                       keep_cnt, decode_out + 1, (long)out_elem, kRfDet, keep_det);
    return trtx::check_launch("trtx_retina_nms");
}

// ---- rpnDecode (RpnDecode.cu:27-143)
extern "C" size_t trtx_rpn_decode_workspace(int batch, int num_anchors, int height, int width) {
    // sized for the worst case top_n >= n (keys of every anchor); with top_n < n only next_pow2(top_n) of them are used
    const size_t n = (size_t)num_anchors * height * width, n_pad = next_pow2((int)n);
    return align_up(batch * n_pad * 8, 256) + align_up(batch * n_pad * 4, 256) + align_up((size_t)num_anchors * 16, 256);
}

extern "C" int32_t trtx_rpn_decode(int batch, const float* scores, const float* deltas, int height, int width,
                                   int image_height, int image_width, float stride, const float* anchors_host,
                                   int num_anchors, int top_n, float* out_scores, float* out_boxes, void* workspace,
                                   size_t workspace_bytes, hipStream_t stream) {
    if (!scores || !deltas || !anchors_host || !out_scores || !out_boxes || !workspace || batch < 1 || top_n < 1)
        return TRTX_ERR_INVALID;
    const int n = num_anchors * height * width;
    // keys that take part in the sort: the top_n selected ones (radix select), or all of them when nothing is dropped
    const int n_pad = n > top_n ? next_pow2(top_n) : next_pow2(n);
    Carver c{static_cast<char*>(workspace), workspace_bytes};
    uint64_t* keys = c.take<uint64_t>((size_t)batch * n_pad);
    int* order = c.take<int>((size_t)batch * n_pad);
    float* anchors = c.take<float>((size_t)num_anchors * 4);
    if (!c.ok) return TRTX_ERR_WORKSPACE;
    TRTX_HIP_TRY(hipMemcpyAsync(anchors, anchors_host, sizeof(float) * num_anchors * 4, hipMemcpyHostToDevice, stream));
    int num = n;
    const int* order_arg = nullptr;
    if (n > top_n) {
        hipLaunchKernelGGL(topk_select_kernel, dim3(batch), dim3(1024), 0, stream, scores, n, top_n, n_pad, keys);
        const int32_t st = sort_keys(keys, batch, n_pad, stream);
        if (st != TRTX_OK) return st;
        hipLaunchKernelGGL(unpack_keys_kernel, grid1(n_pad, batch), dim3(256), 0, stream, keys, n_pad, scores, n, n_pad, order,
                           (float*)nullptr);
        order_arg = order;
        num = top_n;
    }
    hipLaunchKernelGGL(rpn_decode_kernel, grid1(top_n, batch), dim3(256), 0, stream, order_arg, n_pad, num, top_n, scores, deltas, n,
                       height, width, num_anchors, stride, anchors, (float)image_width, (float)image_height, out_scores,
                       out_boxes);
    return trtx::check_launch("trtx_rpn_decode");
}

// ---- shared: sort -> greedy NMS -> stable re-sort -> gather
static int32_t sorted_nms(int mode, int batch, const float* scores, const float* boxes, const float* classes, int n, int n_out,
                          float thresh, float* out_scores, float* out_boxes, float* out_classes, void* workspace,
                          size_t workspace_bytes, hipStream_t stream) {
    const int n_pad = next_pow2(n);
    Carver c{static_cast<char*>(workspace), workspace_bytes};
    uint64_t* keys = c.take<uint64_t>((size_t)batch * n_pad);
    int* order = c.take<int>((size_t)batch * n);
    float* sorted = c.take<float>((size_t)batch * n);
    if (!c.ok) return TRTX_ERR_WORKSPACE;
    hipLaunchKernelGGL(make_keys_kernel, grid1(n_pad, batch), dim3(256), 0, stream, scores, n, n_pad, keys);
    int32_t st = sort_keys(keys, batch, n_pad, stream);
    if (st != TRTX_OK) return st;
    hipLaunchKernelGGL(unpack_keys_kernel, grid1(n, batch), dim3(256), 0, stream, keys, n_pad, scores, n, n, order, sorted);
    NmsArgs a{};
    a.box_base = boxes;
    a.box_batch_stride = (long)n * 4;
    a.box_stride = 4;
    a.classes = classes;
    a.n_src = n;
    a.order = order;
    a.scores = sorted;
    a.n_cap = n;
    a.thresh = thresh;
    a.mode = mode;
    a.stop_after = (mode == NMS_RPN || mode == NMS_HARD0) ? n_out : 0;
    if (mode == NMS_RPN && n <= kMaskMaxN) {
        const int n_blk = (n + 63) / 64;
        uint64_t* mask_t = c.take<uint64_t>((size_t)batch * n_blk * n_blk * 64);
        if (!c.ok) return TRTX_ERR_WORKSPACE;
        hipLaunchKernelGGL(hard_mask_kernel, dim3(n_blk, n_blk, batch), dim3(64), 0, stream, boxes, (long)n * 4, order, n, n_blk,
                           thresh, mask_t);
        hipLaunchKernelGGL(hard_scan_kernel, dim3(batch), dim3(1024), 0, stream, mask_t, sorted, n, n_blk, a.stop_after);
    } else if ((mode == NMS_HARD0 || mode == NMS_SOFT_LINEAR || mode == NMS_SOFT_GAUSS) && classes && n <= 1024) {
        const int n_blk = (n + 63) / 64;
        float* f_t = c.take<float>((size_t)batch * n_blk * 64 * n_blk * 64);
        if (!c.ok) return TRTX_ERR_WORKSPACE;
        hipLaunchKernelGGL(soft_factor_kernel, dim3(n_blk, n_blk, batch), dim3(64), 0, stream, boxes, (long)n * 4, classes, n, order, n, n_blk, thresh,
                           mode, f_t);
        hipLaunchKernelGGL(greedy_nms_factor_kernel, dim3(batch), dim3(1024), 0, stream, sorted, n, n_blk, f_t, mode, a.stop_after);
    } else {
        if (a.n_cap <= 1024) hipLaunchKernelGGL(greedy_nms_small_kernel, dim3(batch), dim3(1024), 0, stream, a);
        else hipLaunchKernelGGL(greedy_nms_kernel, dim3(batch), dim3(1024), 0, stream, a);
    }
    hipLaunchKernelGGL(rekey_kernel, grid1(n_pad, batch), dim3(256), 0, stream, sorted, n, n_pad, keys);
    st = sort_keys(keys, batch, n_pad, stream);
    if (st != TRTX_OK) return st;
    hipLaunchKernelGGL(gather_after_nms_kernel, grid1(n_out, batch), dim3(256), 0, stream, keys, n_pad, order, n, sorted, boxes,
                       classes, n_out, out_scores, out_boxes, out_classes);
    return trtx::check_launch("sorted_nms");
}

extern "C" size_t trtx_sorted_nms_workspace(int batch, int n) {
    const size_t n_pad = next_pow2(n);
    const size_t n_blk = ((size_t)n + 63) / 64;
    const size_t mask = n <= kMaskMaxN ? align_up((size_t)batch * n_blk * n_blk * 64 * 8, 256) : 0;  // hard_mask_kernel
    const size_t factors = n <= 1024 ? align_up((size_t)batch * n_blk * 64 * n_blk * 64 * 4, 256) : 0;   // soft_factor_kernel (BatchedNms)
    return align_up(batch * n_pad * 8, 256) + 2 * align_up((size_t)batch * n * 4, 256) + (mask > factors ? mask : factors);
}

// rpnNms (RpnNms.cu:59-121)
extern "C" int32_t trtx_rpn_nms(int batch, const float* scores, const float* boxes, int pre_nms_topk, int post_nms_topk,
                                float nms_thresh, float* out_boxes, void* workspace, size_t workspace_bytes,
                                hipStream_t stream) {
    if (!scores || !boxes || !out_boxes || !workspace || batch < 1 || pre_nms_topk < 1 || post_nms_topk < 1) return TRTX_ERR_INVALID;
    return sorted_nms(NMS_RPN, batch, scores, boxes, nullptr, pre_nms_topk, post_nms_topk, nms_thresh, nullptr, out_boxes, nullptr,
                      workspace, workspace_bytes, stream);
}

// batchedNms (BatchedNms.cu:90-162); method 0 hard, 1 soft-linear, 2 soft-gaussian
extern "C" int32_t trtx_batched_nms(int nms_method, int batch, const float* scores, const float* boxes, const float* classes,
                                    int count, int detections_per_im, float nms_thresh, float* out_scores, float* out_boxes,
                                    float* out_classes, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!scores || !boxes || !classes || !out_scores || !out_boxes || !out_classes || !workspace || batch < 1 || count < 1)
        return TRTX_ERR_INVALID;
    const int mode = nms_method == 1 ? NMS_SOFT_LINEAR : (nms_method == 2 ? NMS_SOFT_GAUSS : NMS_HARD0);
    return sorted_nms(mode, batch, scores, boxes, classes, count, detections_per_im, nms_thresh, out_scores, out_boxes, out_classes,
                      workspace, workspace_bytes, stream);
}

// roiAlign (RoiAlign.cu:155-182)
// predictorDecode (PredictorDecode.cu:24-110)
extern "C" size_t trtx_predictor_decode_workspace(int batch, int num_boxes, int num_classes) {
    const size_t n_pad = next_pow2(num_boxes * num_classes);
    return align_up(batch * n_pad * 8, 256) + align_up(batch * n_pad * 4, 256);
}

extern "C" int32_t trtx_predictor_decode(int batch, const float* scores, const float* deltas, const float* proposals,
                                         int num_boxes, int num_classes, int image_height, int image_width,
                                         const float* bbox_reg_weights_host, float* out_scores, float* out_boxes,
                                         float* out_classes, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    (void)image_height;
    if (!scores || !deltas || !proposals || !bbox_reg_weights_host || !out_scores || !out_boxes || !out_classes || !workspace ||
        batch < 1)
        return TRTX_ERR_INVALID;
    const int n = num_boxes * num_classes;
    // only the best num_boxes of the num_boxes * num_classes (box, class) scores survive (PredictorDecode.cu:65-78 sorts all of
    // them): radix select first, sort just the survivors
    const bool select = n > num_boxes && num_classes > 1;
    const int n_pad = select ? next_pow2(num_boxes) : next_pow2(n);
    Carver c{static_cast<char*>(workspace), workspace_bytes};
    uint64_t* keys = c.take<uint64_t>((size_t)batch * n_pad);
    int* order = c.take<int>((size_t)batch * n_pad);
    if (!c.ok) return TRTX_ERR_WORKSPACE;
    if (select)
        hipLaunchKernelGGL(topk_select_kernel, dim3(batch), dim3(1024), 0, stream, scores, n, num_boxes, n_pad, keys);
    else
        hipLaunchKernelGGL(make_keys_kernel, grid1(n_pad, batch), dim3(256), 0, stream, scores, n, n_pad, keys);
    const int32_t st = sort_keys(keys, batch, n_pad, stream);
    if (st != TRTX_OK) return st;
    hipLaunchKernelGGL(unpack_keys_kernel, grid1(n_pad, batch), dim3(256), 0, stream, keys, n_pad, scores, n, n_pad, order,
                       (float*)nullptr);
    const float* w = bbox_reg_weights_host;
    hipLaunchKernelGGL(predictor_decode_kernel, grid1(num_boxes, batch), dim3(256), 0, stream, order, n_pad, scores, deltas, proposals,
                       num_boxes, num_classes, (float)image_width, w[0], w[1], w[2], w[3], out_scores, out_boxes, out_classes);
    return trtx::check_launch("trtx_predictor_decode");
}

// ---- maskRcnnInference (MaskRcnnInference.cu:8-62): per detection, the mask plane of its predicted class through a sigmoid.
// labels [batch][D] (class ids as floats), masks [batch][D][C][S][S] -> out [batch][D][1][S][S].  The reference leaves the
// output untouched when the class id is outside [0, C); here such planes are written as zeros.
__global__ void mask_select_kernel(const float* __restrict__ labels, const float* __restrict__ masks, int detections,
                                   int plane, int num_classes, long total, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long det = i / plane;  // detection index across the batch
    const int px = (int)(i - det * plane);
    const int cls = (int)labels[det];
    float v = 0.0f;
    if (cls >= 0 && cls < num_classes) v = 1.0f / (1.0f + expf(-masks[(det * num_classes + cls) * plane + px]));
    out[i] = v;
}

extern "C" int32_t trtx_mask_rcnn_inference(int batch, const float* labels, const float* masks, int detections_per_im,
                                            int output_size, int num_classes, float* out_masks, hipStream_t stream) {
    if (!labels || !masks || !out_masks || batch < 1 || detections_per_im < 1 || output_size < 1 || num_classes < 1)
        return TRTX_ERR_INVALID;
    const int plane = output_size * output_size;
    const long total = (long)batch * detections_per_im * plane;
    hipLaunchKernelGGL(mask_select_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, labels, masks,
                       detections_per_im, plane, num_classes, total, out_masks);
    return trtx::check_launch("trtx_mask_rcnn_inference");
}
