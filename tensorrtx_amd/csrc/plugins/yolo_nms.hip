// Class-aware greedy NMS for the YOLOv8 decode buffer — one workgroup (16 waves) per image.
//
// Replaces the reference's single-threaded host nms()/batch_nms() (yolov8/src/postprocess.cpp:71-129):
//   keep candidates with conf > conf_thresh (NaN dropped), group by class ascending, order by
//   conf descending then bbox[0] ascending, greedy-suppress later boxes with iou > nms_thresh.
// Selection is bit-exact with the sequential algorithm: IoU uses the same IEEE operations in the
// same order (this file is compiled with -ffp-contract=off), ties of (class, conf, bbox[0]) are
// broken by decode slot index (the reference's std::sort leaves them unspecified).
//
// Structure (wave64):
//   1. 128-bit composite keys (class | ~conf | bbox[0] | slot) sorted by a 1024-wide bitonic network:
//      strides < 64 are exchanged with wave shuffles, strides >= 64 through LDS (10 of 55 stages).
//   2. blocked greedy pass: block bi (64 sorted boxes = wave bi) is resolved inside one wave with a
//      64x64 suppression bit-matrix (one row per lane) and a scalar ballot chain; its surviving boxes
//      are then applied by every later wave to its own boxes.  One barrier per block.
//   3. ordered compaction of the survivors (ballot + popcount scan).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"

namespace {

constexpr int kCap = 1024;  // candidates per image held in LDS (reference: kMaxNumOutputBbox = 1000)

__device__ __forceinline__ bool key_less(uint64_t ah, uint64_t al, uint64_t bh, uint64_t bl) {
    return ah < bh || (ah == bh && al < bl);
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    const uint32_t lo = __shfl_xor((int)(uint32_t)v, m);
    const uint32_t hi = __shfl_xor((int)(uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}

// postprocess.cpp:71-85, operation for operation.
__device__ __forceinline__ float iou_xyxy(const float4 l, const float4 r) {
    const float ib0 = l.x < r.x ? r.x : l.x;  // max(l[0], r[0])
    const float ib1 = r.z < l.z ? r.z : l.z;  // min(l[2], r[2])
    const float ib2 = l.y < r.y ? r.y : l.y;  // max(l[1], r[1])
    const float ib3 = r.w < l.w ? r.w : l.w;  // min(l[3], r[3])
    if (ib2 > ib3 || ib0 > ib1) return 0.0f;
    const float inter = (ib1 - ib0) * (ib3 - ib2);
    const float uni = (l.z - l.x) * (l.w - l.y) + (r.z - r.x) * (r.w - r.y) - inter;
    return inter / uni;
}

__global__ __launch_bounds__(kCap) void yolo_nms_kernel(const float* __restrict__ decode, int out_elem,
                                                        int det_floats, int max_out, float conf_thresh,
                                                        float nms_thresh, int* __restrict__ keep_idx,
                                                        int* __restrict__ keep_cnt, float* __restrict__ keep_det, long long* __restrict__ dbg) {
    __shared__ uint64_t s_hi[kCap];
    __shared__ uint64_t s_lo[kCap];
    __shared__ float4 s_box[kCap];
    __shared__ float s_cls[kCap];
    __shared__ float s_conf[kCap];
    __shared__ uint64_t s_kept[kCap / 64];
    __shared__ int s_wcnt[kCap / 64];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* img = decode + (size_t)b * out_elem;
    int count = (int)img[0];
    count = count < max_out ? count : max_out;
    count = count < kCap ? count : kCap;

    long long t0 = 0;
    if (dbg) t0 = wall_clock64();
#define TRTX_NMS_STAMP(i) if (dbg && tid == 0 && b == 0) dbg[i] = wall_clock64() - t0
    // ---- load + keys ------------------------------------------------------------------------
    uint64_t hi = ~0ull, lo = ~0ull;
    if (tid < count) {
        const float* det = img + 1 + (size_t)tid * det_floats;
        const float4 box = make_float4(det[0], det[1], det[2], det[3]);
        const float conf = det[4];
        const float cls = det[5];
        s_box[tid] = box;
        s_cls[tid] = cls;
        s_conf[tid] = conf;
        if (conf > conf_thresh) {  // false for NaN, as "conf <= thresh || isnan" drops (postprocess.cpp:99)
            hi = ((uint64_t)trtx::ord_f32(cls) << 32) | (uint32_t)~trtx::ord_f32(conf);
            lo = ((uint64_t)trtx::ord_f32(box.x) << 32) | (uint32_t)tid;
        }
    }

    TRTX_NMS_STAMP(0);
    // ---- bitonic sort, ascending, 1024 keys ---------------------------------------------------
    for (int k = 2; k <= kCap; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            uint64_t ph, pl;
            if (j >= 64) {
                s_hi[tid] = hi;
                s_lo[tid] = lo;
                __syncthreads();
                ph = s_hi[tid ^ j];
                pl = s_lo[tid ^ j];
                __syncthreads();
            } else {
                ph = shfl_xor_u64(hi, j);
                pl = shfl_xor_u64(lo, j);
            }
            const bool up = (tid & k) == 0;
            const bool lower = (tid & j) == 0;
            const bool take_min = (up == lower);
            const bool swap = take_min ? key_less(ph, pl, hi, lo) : key_less(hi, lo, ph, pl);
            if (swap) {
                hi = ph;
                lo = pl;
            }
        }
    }
    __syncthreads();

    TRTX_NMS_STAMP(1);
    // ---- gather the sorted records --------------------------------------------------------------
    const bool valid = !(hi == ~0ull && lo == ~0ull);
    const int orig = valid ? (int)(uint32_t)lo : 0;
    float4 my_box = make_float4(0.f, 0.f, 0.f, 0.f);
    float my_cls = -1.0f, my_conf = 0.0f;
    if (valid) {
        my_box = s_box[orig];
        my_cls = s_cls[orig];
        my_conf = s_conf[orig];
    }
    {
        const unsigned long long m = __ballot(valid);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
    }
    __syncthreads();
    int n = 0;
#pragma unroll
    for (int w = 0; w < kCap / 64; ++w) n += s_wcnt[w];
    s_box[tid] = my_box;  // now indexed by sorted rank
    s_cls[tid] = my_cls;
    __syncthreads();

    TRTX_NMS_STAMP(2);
    // ---- blocked greedy suppression -------------------------------------------------------------
    bool rem = !valid;
    const int nblk = (n + 63) >> 6;
    for (int bi = 0; bi < nblk; ++bi) {
        if (wave == bi) {
            // Keys are sorted by class first, so the boxes that can suppress this lane's box are the contiguous run of
            // equal-class lanes right before it: walk that run only (typically a handful of boxes, not 63).
            uint64_t supby = 0;  // bit k: sorted box (64*bi + k), k < lane, would suppress this lane's box
            for (int k = lane - 1; k >= 0; --k) {
                const int rk = (bi << 6) + k;
                if (s_cls[rk] != my_cls) break;
                if (iou_xyxy(s_box[rk], my_box) > nms_thresh) supby |= 1ull << k;
            }
            uint64_t dead = __ballot(rem);
            for (int k = 0; k < 64; ++k) {
                if (!((dead >> k) & 1ull)) dead |= __ballot((supby >> k) & 1ull);
            }
            rem = (dead >> lane) & 1ull;
            if (lane == 0) s_kept[bi] = ~dead;
        }
        __syncthreads();
        // Later boxes have a class >= every class of block bi, so only the tail run of block bi with exactly this
        // class can matter: find its start by binary search (classes are sorted) and test the KEPT boxes of that run.
        if (wave > bi && !rem && s_cls[(bi << 6) + 63] == my_cls) {
            int lo = 0, hi = 63;  // first index in the block whose class equals my_cls
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_cls[(bi << 6) + mid] < my_cls) lo = mid + 1; else hi = mid;
            }
            uint64_t kept = s_kept[bi] & (~0ull << lo);
            while (kept) {
                const int k = __ffsll((unsigned long long)kept) - 1;
                kept &= kept - 1;
                if (iou_xyxy(s_box[(bi << 6) + k], my_box) > nms_thresh) {
                    rem = true;
                    break;
                }
            }
        }
    }

    TRTX_NMS_STAMP(3);
    // ---- ordered compaction -----------------------------------------------------------------------
    const bool keep = valid && !rem;
    const unsigned long long km = __ballot(keep);
    __syncthreads();  // s_wcnt reuse
    if (lane == 0) s_wcnt[wave] = __popcll(km);
    __syncthreads();
    int pos = __popcll(km & ((1ull << lane) - 1ull));
    int total = 0;
#pragma unroll
    for (int w = 0; w < kCap / 64; ++w) {
        const int c = s_wcnt[w];
        if (w < wave) pos += c;
        total += c;
    }
    if (keep) {
        keep_idx[(size_t)b * max_out + pos] = orig;
        if (keep_det) {
            float* o = keep_det + ((size_t)b * max_out + pos) * 6;
            o[0] = my_box.x;
            o[1] = my_box.y;
            o[2] = my_box.z;
            o[3] = my_box.w;
            o[4] = my_conf;
            o[5] = my_cls;
        }
    }
    if (tid == 0) keep_cnt[b] = total;
    TRTX_NMS_STAMP(4);
    if (dbg && tid == 0 && b == 0) dbg[5] = n;
}

}  // namespace

extern "C" int32_t trtx_yolo_nms(const float* decode_out, int batch, int max_out, float conf_thresh,
                                 float nms_thresh, int32_t* keep_idx, int32_t* keep_cnt, float* keep_det,
                                 hipStream_t stream) {
    if (!decode_out || !keep_idx || !keep_cnt || batch < 1 || max_out < 1) return TRTX_ERR_INVALID;
    if (max_out > kCap) return TRTX_ERR_UNSUPPORTED;
    const int out_elem = 1 + max_out * trtx::kYoloDetFloats;
    hipLaunchKernelGGL(yolo_nms_kernel, dim3(batch), dim3(kCap), 0, stream, decode_out, out_elem,
                       trtx::kYoloDetFloats, max_out, conf_thresh, nms_thresh, keep_idx, keep_cnt, keep_det, (long long*)nullptr);
    return trtx::check_launch("trtx_yolo_nms");
}

// development probe: same kernel with per-phase wall-clock stamps of image 0 (100 MHz ticks) in dbg[0..5]
extern "C" int32_t trtx_yolo_nms_probe(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh,
                                       int32_t* keep_idx, int32_t* keep_cnt, float* keep_det, long long* dbg, hipStream_t stream) {
    const int out_elem = 1 + max_out * trtx::kYoloDetFloats;
    hipLaunchKernelGGL(yolo_nms_kernel, dim3(batch), dim3(kCap), 0, stream, decode_out, out_elem, trtx::kYoloDetFloats, max_out,
                       conf_thresh, nms_thresh, keep_idx, keep_cnt, keep_det, dbg);
    return trtx::check_launch("trtx_yolo_nms_probe");
}
