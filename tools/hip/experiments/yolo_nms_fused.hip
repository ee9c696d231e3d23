// Class-aware greedy NMS for the YOLOv8 decode buffer.
//
// Replaces the reference's single-threaded host nms()/batch_nms() (yolov8/src/postprocess.cpp:71-129):
//   keep candidates with conf > conf_thresh (NaN dropped), group by class ascending, order by
//   conf descending then bbox[0] ascending, greedy-suppress later boxes with iou > nms_thresh.
// Selection is bit-exact with the sequential algorithm: IoU uses the same IEEE operations in the
// same order (this file is compiled with -ffp-contract=off), ties of (class, conf, bbox[0]) are
// broken by decode slot index (the reference's std::sort leaves them unspecified).
//
// One launch (wave64), one workgroup per image: sort -> suppression rows in registers -> turn-taking scan -> ordered compaction (yolo_nms_fused_kernel below).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

#ifndef TRTX_NMS_STAMP   // phase stamps of the fused kernel (tools/hip/nms_anatomy.hip)
#define TRTX_NMS_STAMP(i)
#endif

namespace {

constexpr int kCap = 1024;        // candidates per image (reference: kMaxNumOutputBbox = 1000)
constexpr int kBlocks = kCap / 64;

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

// yolov5/src/postprocess.cpp:30-44: centre-format boxes (cx, cy, w, h), operation for operation.
__device__ __forceinline__ float iou_cxcywh(const float4 l, const float4 r) {
    const float a0 = l.x - l.z / 2.f, b0 = r.x - r.z / 2.f;
    const float a1 = l.x + l.z / 2.f, b1 = r.x + r.z / 2.f;
    const float a2 = l.y - l.w / 2.f, b2 = r.y - r.w / 2.f;
    const float a3 = l.y + l.w / 2.f, b3 = r.y + r.w / 2.f;
    const float ib0 = a0 < b0 ? b0 : a0;  // max
    const float ib1 = b1 < a1 ? b1 : a1;  // min
    const float ib2 = a2 < b2 ? b2 : a2;
    const float ib3 = b3 < a3 ? b3 : a3;
    if (ib2 > ib3 || ib0 > ib1) return 0.0f;
    const float inter = (ib1 - ib0) * (ib3 - ib2);
    return inter / (l.z * l.w + r.z * r.w - inter);
}

// probiou() of the host NMS for oriented boxes (yolov8/src/postprocess.cpp:303-355): the reference mixes float and double
// (std::pow(float, int) and the 12.0 / 1.0 literals promote to double, std::cos/sin/sqrt/exp(float) stay float); the same
// promotions are written out here.  pow(x, 2) is x * x in double (exact up to the final rounding either way).
__device__ __forceinline__ void obb_cov(const float4 box, float angle, float& a_val, float& b_val, float& c_val) {
    const float w = box.z, h = box.w;
    const float a = (float)(w * w / 12.0);
    const float b = (float)(h * h / 12.0);
    const float cos_r = cosf(angle), sin_r = sinf(angle);
    const float cos_r2 = cos_r * cos_r, sin_r2 = sin_r * sin_r;
    a_val = a * cos_r2 + b * sin_r2;
    b_val = a * sin_r2 + b * cos_r2;
    c_val = (a - b) * cos_r * sin_r;
}
__device__ __forceinline__ float probiou_host(const float4 r1, float ang1, const float4 r2, float ang2) {
    const float eps = 1e-7f;
    float a1, b1, c1, a2, b2, c2;
    obb_cov(r1, ang1, a1, b1, c1);
    obb_cov(r2, ang2, a2, b2, c2);
    const float x1 = r1.x, y1 = r1.y, x2 = r2.x, y2 = r2.y;
    const double dy = (double)(y1 - y2), dx = (double)(x1 - x2), cc = (double)(c1 + c2);
    const double den = (double)((a1 + a2) * (b1 + b2)) - cc * cc + (double)eps;
    const float t1 = (float)(((double)(a1 + a2) * (dy * dy) + (double)(b1 + b2) * (dx * dx)) / den);
    const float t2 = (float)((double)((c1 + c2) * (x2 - x1) * (y1 - y2)) / den);
    const float s1 = a1 * b1 - c1 * c1, s2 = a2 * b2 - c2 * c2;
    const float den3 = 4 * sqrtf(s1 > 0.0f ? s1 : 0.0f) * sqrtf(s2 > 0.0f ? s2 : 0.0f) + eps;
    const float t3 = (float)log(((double)((a1 + a2) * (b1 + b2)) - cc * cc) / (double)den3 + (double)eps);
    float bd = 0.25f * t1 + 0.5f * t2 + 0.5f * t3;
    bd = bd < 100.0f ? bd : 100.0f;
    bd = bd > eps ? bd : eps;
    const float hd = (float)sqrt(1.0 - (double)expf(-bd) + (double)eps);
    return 1 - hd;
}

// The workspace the entry points ask for (trtx_yolo_nms_workspace): rounds 1-5 ran three launches that handed sorted records and a 128 KB suppression matrix per
// image through it; the fused kernel below keeps everything on chip.  The size stays (callers allocate by it), the bytes are not touched.
__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline size_t nms_ws_bytes(int batch) {
    const size_t rec = (size_t)batch * kCap;
    return align256(rec * 16) + 4 * align256(rec * 4) + align256((size_t)batch * 4) + align256(rec * kBlocks * sizeof(uint64_t));
}

// MODE 0: YOLOv8 host nms() (postprocess.cpp:94-121): xyxy boxes, "conf <= thresh || isnan" dropped, ties by bbox[0] then slot.
// MODE 1: YOLOv5 host nms() (yolov5/src/postprocess.cpp:50-73): centre-format boxes, "conf <= thresh" dropped (a NaN stays),
//         cmp() orders by conf only -> ties by slot (the reference's unstable std::sort leaves them unspecified).
// MODE 2: YOLOv8 host nms_obb() (postprocess.cpp:357-385): (cx, cy, w, h, angle) boxes, "conf <= thresh" dropped, same cmp() as
//         MODE 0, a later box is erased when probiou(item, box) >= thresh.
//
// ONE launch, one workgroup (16 waves) per image (round 6; rounds 1-5: sort / mask / scan as three launches, 53.5 us of a YOLOv8n b32 step for ~700 candidates
// per image - latency, not work: profiles/r05_kernel_stats_c3_1ctx_lanes1.txt):
//   1. 128-bit composite keys (class | ~conf | bbox[0] | slot) through a 1024-wide bitonic network (strides < 64 by wave shuffles, >= 64 through LDS);
//   2. the sorted boxes and classes go to LDS; every lane builds the suppression word of its box against the earlier boxes of ITS OWN 64-box block;
//   3. the waves take turns in block order: the block's in-block chain, then every later wave tests its boxes against the boxes that block KEPT - a greedy NMS
//      only ever compares with kept boxes; ordered compaction of the kept records.
template <int MODE>
__global__ __launch_bounds__(kCap) void yolo_nms_fused_kernel(const float* __restrict__ decode, int out_elem, int det_floats, int max_out, float conf_thresh,
                                                              float nms_thresh, int* __restrict__ keep_idx, int* __restrict__ keep_cnt,
                                                              float* __restrict__ keep_det, int det_out) {
    __shared__ uint64_t s_hi[kCap];
    __shared__ uint64_t s_lo[kCap];
    __shared__ float4 s_box[kCap];
    __shared__ float s_cls[kCap];
    __shared__ float s_ang[MODE == 2 ? kCap : 1];
    __shared__ uint64_t s_kept[kBlocks];
    __shared__ int s_wcnt[kBlocks];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* img = decode + (size_t)b * out_elem;
    int count = (int)img[0];
    count = count < max_out ? count : max_out;
    count = count < kCap ? count : kCap;

    TRTX_NMS_STAMP(0);
    uint64_t hi = ~0ull, lo = ~0ull;
    if (tid < count) {
        const float* det = img + 1 + (size_t)tid * det_floats;
        const float conf = det[4];
        // MODE 0: false for NaN, as "conf <= thresh || isnan" drops (postprocess.cpp:99); MODE 1: only "conf <= thresh" drops
        if (MODE == 0 ? (conf > conf_thresh) : !(conf <= conf_thresh)) {
            hi = ((uint64_t)trtx::ord_f32(det[5]) << 32) | (uint32_t)~trtx::ord_f32(conf);
            lo = ((uint64_t)(MODE != 1 ? trtx::ord_f32(det[0]) : 0u) << 32) | (uint32_t)tid;
        }
    }
    TRTX_NMS_STAMP(1);
    // bitonic sort, ascending, 1024 keys
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
    TRTX_NMS_STAMP(2);
    // this lane's sorted record (the decode buffer is re-read through L2: 24 B per record)
    const bool valid = !(hi == ~0ull && lo == ~0ull);
    const int orig = valid ? (int)(uint32_t)lo : 0;
    float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
    float cls = -1.0f, conf = 0.0f, ang = 0.0f;
    if (valid) {
        const float* det = img + 1 + (size_t)orig * det_floats;
        box = make_float4(det[0], det[1], det[2], det[3]);
        conf = det[4];
        cls = det[5];
        if (MODE == 2) ang = det[det_floats - 1];  // Detection::angle is the last float of the record
    }
    s_box[tid] = box;
    s_cls[tid] = cls;
    if (MODE == 2) s_ang[tid] = ang;
    {
        const unsigned long long m = __ballot(valid);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
    }
    __syncthreads();
    TRTX_NMS_STAMP(3);
    int n = 0;
#pragma unroll
    for (int w = 0; w < kBlocks; ++w) n += s_wcnt[w];

    // ---- in-block suppression words, all blocks at once: bit k of `supby` says "sorted box 64 wave + k suppresses this lane's box".  Valid records come first
    // (their keys are below the padding's) and classes ascend, so the boxes that can suppress box i are the run [first box of i's class, i): its start is the
    // first j with ord(class j) >= ord(class i) - a lower bound over the sorted prefix.
    int seg = tid;
    uint64_t supby = 0;
    if (valid) {
        const uint32_t mykey = trtx::ord_f32(cls);
        int lo_i = 0, hi_i = tid;
        while (lo_i < hi_i) {
            const int mid = (lo_i + hi_i) >> 1;
            if (trtx::ord_f32(s_cls[mid]) < mykey) lo_i = mid + 1;
            else hi_i = mid;
        }
        seg = lo_i;
        for (int k = seg > 64 * wave ? seg : 64 * wave; k < tid; ++k) {   // only earlier boxes suppress
            if (s_cls[k] != cls) continue;
            bool hit;
            if (MODE == 0) hit = iou_xyxy(s_box[k], box) > nms_thresh;
            else if (MODE == 1) hit = iou_cxcywh(s_box[k], box) > nms_thresh;
            else hit = probiou_host(s_box[k], s_ang[k], box, ang) >= nms_thresh;
            if (hit) supby |= 1ull << (k - 64 * wave);
        }
    }
    // columns of this block's own 64x64 tile that suppress anything at all (wave-wide OR of the rows)
    uint64_t colany = supby;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) colany |= shfl_xor_u64(colany, m);

    // ---- the blocks in order.  Wave bi settles its block: its lanes already know whether a KEPT box of an earlier block suppresses them (below); the in-block
    // chain visits, in ascending order, the live boxes that suppress something - exactly the sequential greedy order.  It publishes the kept boxes of the block,
    // and every LATER wave tests its boxes against just those (a greedy NMS only ever compares with kept boxes: ~30 per image instead of ~700 candidates; the
    // three-launch version of rounds 1-5 built the whole lower-triangular matrix, 240k IoU tests for 700 boxes of one class, spread over the chip).
    TRTX_NMS_STAMP(4);
    bool rem = !valid;
    const int nblk = (n + 63) >> 6;
    for (int bi = 0; bi < nblk; ++bi) {
        if (wave == bi) {
            uint64_t dead = __ballot(rem);
            uint64_t todo = colany & ~dead;
            while (todo) {
                const int k = __ffsll((unsigned long long)todo) - 1;
                const uint64_t col = __ballot((supby >> k) & 1ull);
                dead |= col;
                todo &= ~col & ~(1ull << k);
            }
            rem = (dead >> lane) & 1ull;
            if (lane == 0) s_kept[bi] = ~dead;
        }
        __syncthreads();
        if (wave > bi && valid && !rem && seg < 64 * bi + 64) {   // (a class run that starts beyond block bi has nothing in it)
            uint64_t kept = s_kept[bi];
            const int kbeg = seg > 64 * bi ? seg - 64 * bi : 0;   // bits below the run's start belong to other classes
            kept = kbeg > 0 ? (kept >> kbeg) << kbeg : kept;
            while (kept) {
                const int kk = __ffsll((unsigned long long)kept) - 1;
                kept &= kept - 1;
                const int k = 64 * bi + kk;
                if (s_cls[k] != cls) continue;
                bool hit;
                if (MODE == 0) hit = iou_xyxy(s_box[k], box) > nms_thresh;
                else if (MODE == 1) hit = iou_cxcywh(s_box[k], box) > nms_thresh;
                else hit = probiou_host(s_box[k], s_ang[k], box, ang) >= nms_thresh;
                if (hit) {
                    rem = true;
                    break;
                }
            }
        }
    }
    TRTX_NMS_STAMP(5);
    // ordered compaction
    const bool keep = valid && !rem;
    const unsigned long long km = __ballot(keep);
    __syncthreads();   // (s_wcnt is reused: every wave has read the valid counts)
    if (lane == 0) s_wcnt[wave] = __popcll(km);
    __syncthreads();
    int pos = __popcll(km & ((1ull << lane) - 1ull));
    int total = 0;
#pragma unroll
    for (int w = 0; w < kBlocks; ++w) {
        const int cnt = s_wcnt[w];
        if (w < wave) pos += cnt;
        total += cnt;
    }
    if (keep) {
        keep_idx[(size_t)b * max_out + pos] = orig;
        if (keep_det) {
            float* o = keep_det + ((size_t)b * max_out + pos) * det_out;
            if (det_out > 6) o[6] = ang;
            o[0] = box.x;
            o[1] = box.y;
            o[2] = box.z;
            o[3] = box.w;
            o[4] = conf;
            o[5] = cls;
        }
    }
    if (tid == 0) keep_cnt[b] = total;
    TRTX_NMS_STAMP(6);
}

// ---- the reference's optional GPU post-processing mode "g" (yolov8/src/postprocess.cu:42-111): threshold + copy into
// 7-float records, then the NON-greedy NMS of nms_kernel (a box is dropped if any same-class box with higher confidence -
// equal confidence: higher index - overlaps it, whether or not that box survives).  One workgroup per image, records in LDS.
// Slots keep the input order (the reference assigns them with an atomicAdd, i.e. arbitrarily); slots of positions below the
// threshold stay zero and out[0] = input count, as in the reference.
__device__ __forceinline__ float box_iou_g(const float* a, const float* b) {  // postprocess.cu:73-85
    const float cleft = a[0] > b[0] ? a[0] : b[0];
    const float ctop = a[1] > b[1] ? a[1] : b[1];
    const float cright = a[2] < b[2] ? a[2] : b[2];
    const float cbottom = a[3] < b[3] ? a[3] : b[3];
    float cw = cright - cleft, ch = cbottom - ctop;
    cw = cw > 0.0f ? cw : 0.0f;
    ch = ch > 0.0f ? ch : 0.0f;
    const float c_area = cw * ch;
    if (c_area == 0.0f) return 0.0f;
    float aw = a[2] - a[0], ah = a[3] - a[1], bw = b[2] - b[0], bh = b[3] - b[1];
    aw = aw > 0.0f ? aw : 0.0f;
    ah = ah > 0.0f ? ah : 0.0f;
    bw = bw > 0.0f ? bw : 0.0f;
    bh = bh > 0.0f ? bh : 0.0f;
    return c_area / (aw * ah + bw * bh - c_area);
}

__global__ __launch_bounds__(kCap) void yolo_gpu_post_kernel(const float* __restrict__ decode, int in_elem, int det_floats,
                                                             int max_out, float conf_thresh, float nms_thresh,
                                                             float* __restrict__ out) {
    __shared__ float s_rec[kCap][7];  // odd stride: conflict-free row reads
    const int b = blockIdx.x, p = threadIdx.x;
    const float* img = decode + (size_t)b * in_elem;
    float* dst = out + (size_t)b * (1 + (size_t)max_out * 7);
    int count = (int)img[0];
    count = count < max_out ? count : max_out;
    float rec[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p < count) {
        const float* it = img + 1 + (size_t)p * det_floats;
        if (!(it[4] < conf_thresh)) {
            rec[0] = it[0]; rec[1] = it[1]; rec[2] = it[2]; rec[3] = it[3]; rec[4] = it[4]; rec[5] = it[5]; rec[6] = 1.0f;
        }
    }
#pragma unroll
    for (int e = 0; e < 7; ++e) s_rec[p][e] = rec[e];
    __syncthreads();
    if (p < count) {
        for (int i = 0; i < count; ++i) {
            if (i == p || rec[5] != s_rec[i][5]) continue;
            const float ci = s_rec[i][4];
            if (ci >= rec[4]) {
                if (ci == rec[4] && i < p) continue;
                if (box_iou_g(rec, s_rec[i]) > nms_thresh) {
                    rec[6] = 0.0f;
                    break;
                }
            }
        }
    }
    if (p == 0) dst[0] = (float)count;
    if (p < max_out) {
#pragma unroll
        for (int e = 0; e < 7; ++e) dst[1 + (size_t)p * 7 + e] = rec[e];
    }
}

// ---- the reference's GPU post-processing for oriented boxes (postprocess.cu:7-40 decode_kernel_obb, :113-166 box_probiou /
// nms_kernel_obb; call sites yolov8_obb.cpp): 8-float records cx, cy, w, h, conf, class, keep, angle; float math throughout.
__device__ __forceinline__ void cov_g(float w, float h, float r, float& a, float& b, float& c) {  // postprocess.cu:113-122
    const float a_val = w * w / 12.0f, b_val = h * h / 12.0f;
    const float cos_r = cosf(r), sin_r = sinf(r);
    a = a_val * cos_r * cos_r + b_val * sin_r * sin_r;
    b = a_val * sin_r * sin_r + b_val * cos_r * cos_r;
    c = (a_val - b_val) * sin_r * cos_r;
}
__device__ __forceinline__ float probiou_g(const float* p, const float* q) {  // postprocess.cu:124-142; records: [0..3] box, [7] angle
    const float eps = 1e-7f;
    float a1, b1, c1, a2, b2, c2;
    cov_g(p[2], p[3], p[7], a1, b1, c1);
    cov_g(q[2], q[3], q[7], a2, b2, c2);
    const float cx1 = p[0], cy1 = p[1], cx2 = q[0], cy2 = q[1];
    const float t1 = ((a1 + a2) * powf(cy1 - cy2, 2) + (b1 + b2) * powf(cx1 - cx2, 2)) / ((a1 + a2) * (b1 + b2) - powf(c1 + c2, 2) + eps);
    const float t2 = ((c1 + c2) * (cx2 - cx1) * (cy1 - cy2)) / ((a1 + a2) * (b1 + b2) - powf(c1 + c2, 2) + eps);
    const float t3 = logf(((a1 + a2) * (b1 + b2) - powf(c1 + c2, 2)) /
                                  (4 * sqrtf(fmaxf(a1 * b1 - c1 * c1, 0.0f)) * sqrtf(fmaxf(a2 * b2 - c2 * c2, 0.0f)) + eps) +
                          eps);
    float bd = 0.25f * t1 + 0.5f * t2 + 0.5f * t3;
    bd = fmaxf(fminf(bd, 100.0f), eps);
    const float hd = sqrtf(1.0f - expf(-bd) + eps);
    return 1 - hd;
}

__global__ __launch_bounds__(kCap) void yolo_gpu_post_obb_kernel(const float* __restrict__ decode, int in_elem, int det_floats, int max_out,
                                                                 float conf_thresh, float nms_thresh, float* __restrict__ out) {
    __shared__ float s_rec[kCap][9];  // odd stride
    const int b = blockIdx.x, p = threadIdx.x;
    const float* img = decode + (size_t)b * in_elem;
    float* dst = out + (size_t)b * (1 + (size_t)max_out * 8);
    int count = (int)img[0];
    count = count < max_out ? count : max_out;
    float rec[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p < count) {
        const float* it = img + 1 + (size_t)p * det_floats;
        if (!(it[4] < conf_thresh)) {
            rec[0] = it[0]; rec[1] = it[1]; rec[2] = it[2]; rec[3] = it[3]; rec[4] = it[4]; rec[5] = it[5]; rec[6] = 1.0f;
            rec[7] = it[det_floats - 1];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_rec[p][e] = rec[e];
    __syncthreads();
    if (p < count) {
        for (int i = 0; i < count; ++i) {
            if (i == p || rec[5] != s_rec[i][5]) continue;
            const float ci = s_rec[i][4];
            if (ci >= rec[4]) {
                if (ci == rec[4] && i < p) continue;
                if (probiou_g(rec, s_rec[i]) > nms_thresh) {
                    rec[6] = 0.0f;
                    break;
                }
            }
        }
    }
    if (p == 0) dst[0] = (float)count;
    if (p < max_out) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[1 + (size_t)p * 8 + e] = rec[e];
    }
}

}  // namespace

extern "C" int32_t trtx_yolo_postprocess_gpu(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh,
                                             float* out, hipStream_t stream) {
    if (!decode_out || !out || batch < 1 || max_out < 1) return TRTX_ERR_INVALID;
    if (max_out > kCap) return TRTX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(yolo_gpu_post_kernel, dim3(batch), dim3(kCap), 0, stream, decode_out, 1 + max_out * trtx::kYoloDetFloats,
                       trtx::kYoloDetFloats, max_out, conf_thresh, nms_thresh, out);
    return trtx::check_launch("trtx_yolo_postprocess_gpu");
}

extern "C" size_t trtx_yolo_nms_workspace(int batch) {
    return batch < 1 ? 0 : nms_ws_bytes(batch);
}

namespace {
template <int MODE>
int32_t run_nms(const float* decode_out, int batch, int max_out, int det_floats, float conf_thresh, float nms_thresh, int32_t* keep_idx,
                int32_t* keep_cnt, float* keep_det, void* workspace, size_t workspace_bytes, hipStream_t stream, const char* what) {
    if (!decode_out || !keep_idx || !keep_cnt || batch < 1 || max_out < 1) return TRTX_ERR_INVALID;
    if (max_out > kCap) return TRTX_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < nms_ws_bytes(batch)) return TRTX_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return TRTX_ERR_INVALID;
    const int out_elem = 1 + max_out * det_floats;
    hipLaunchKernelGGL(yolo_nms_fused_kernel<MODE>, dim3(batch), dim3(kCap), 0, stream, decode_out, out_elem, det_floats, max_out, conf_thresh, nms_thresh,
                       keep_idx, keep_cnt, keep_det, MODE == 2 ? 7 : 6);
    return trtx::check_launch(what);
}
}  // namespace

extern "C" int32_t trtx_yolo_nms(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh,
                                 int32_t* keep_idx, int32_t* keep_cnt, float* keep_det, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
    return run_nms<0>(decode_out, batch, max_out, trtx::kYoloDetFloats, conf_thresh, nms_thresh, keep_idx, keep_cnt, keep_det, workspace,
                      workspace_bytes, stream, "trtx_yolo_nms");
}

extern "C" int32_t trtx_yolov5_nms(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh,
                                   int32_t* keep_idx, int32_t* keep_cnt, float* keep_det, void* workspace, size_t workspace_bytes,
                                   hipStream_t stream) {
    return run_nms<1>(decode_out, batch, max_out, 38, conf_thresh, nms_thresh, keep_idx, keep_cnt, keep_det, workspace, workspace_bytes,
                      stream, "trtx_yolov5_nms");
}

// nms_obb / batch_nms_obb (yolov8/src/postprocess.cpp:357-393): oriented boxes, ProbIoU.  keep_det: [batch][max_out][7] =
// cx, cy, w, h, conf, class, angle.
extern "C" int32_t trtx_yolo_nms_obb(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh, int32_t* keep_idx,
                                     int32_t* keep_cnt, float* keep_det, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return run_nms<2>(decode_out, batch, max_out, trtx::kYoloDetFloats, conf_thresh, nms_thresh, keep_idx, keep_cnt, keep_det, workspace,
                      workspace_bytes, stream, "trtx_yolo_nms_obb");
}

// cuda_decode_obb + cuda_nms_obb (yolov8/src/postprocess.cu:7-40, 144-166, 181-193): out [batch][1 + max_out * 8]
extern "C" int32_t trtx_yolo_postprocess_gpu_obb(const float* decode_out, int batch, int max_out, float conf_thresh, float nms_thresh,
                                                 float* out, hipStream_t stream) {
    if (!decode_out || !out || batch < 1 || max_out < 1) return TRTX_ERR_INVALID;
    if (max_out > kCap) return TRTX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(yolo_gpu_post_obb_kernel, dim3(batch), dim3(kCap), 0, stream, decode_out, 1 + max_out * trtx::kYoloDetFloats,
                       trtx::kYoloDetFloats, max_out, conf_thresh, nms_thresh, out);
    return trtx::check_launch("trtx_yolo_postprocess_gpu_obb");
}
