/*
 * ORACLE — test infrastructure only.  Never linked into / called by the product path.
 *
 * CPU restatement (plain C, sequential semantics) of the YOLOv8 detection post-processing
 * of wang-xinyu/tensorrtx:
 *   - decode  : yolov8/plugin/yololayer.cu:178-220 (CalDetection, det branch) and
 *               :282-316 (forwardGpu: per-level launch, outputElem = 1 + maxOut*sizeof(Detection)/4)
 *   - Detection record: yolov8/include/types.h:4-12  (4 bbox + conf + class_id + 32 mask + 51 kpt + 1 angle = 90 floats)
 *   - NMS     : yolov8/src/postprocess.cpp:71-129 (iou, cmp, nms, batch_nms)
 *
 * Parity status: "parity unpinned" — the reference holds no golden vectors for this path
 * (SURVEY.md §8c); this file follows the reference source line by line in behaviour.
 *
 * Two documented canonicalisations (the reference is order-nondeterministic there):
 *   1. decode slot order: the reference hands out slots with atomicAdd (yololayer.cu:206), so the
 *      order of candidates is racy.  Canonical order here = (level, cell) ascending, i.e. the
 *      order a single sequential thread would produce.
 *   2. the slot counter keeps growing past maxOut in the reference (yololayer.cu:206-208) and the
 *      host NMS then reads past the buffer (postprocess.cpp:98).  Here the stored count is
 *      clamped to maxOut (the first maxOut candidates in canonical order are kept).
 *   3. std::sort in nms() is unstable; ties of (conf, bbox[0]) are broken by canonical slot index.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).  -ffp-contract=off
 * keeps every float op a single IEEE operation, the same as the HIP kernels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DET_FLOATS 90 /* sizeof(Detection)/sizeof(float), types.h:4-12 with kNumberOfPoints = 17 */

static float logist(float x) { /* yololayer.cu:174-176 */
    return 1.0f / (1.0f + expf(-x));
}

/* inputs[l] : [batch][4 + classes][grid_h*grid_w] fp32 (CHW, implicit batch outermost)
 * output    : [batch][1 + max_out*90] fp32; output[b][0] = number of candidates (clamped) */
void yolo_decode_ref(const float* const* inputs, int batch, int classes, int net_h, int net_w, const int* strides,
                     int n_strides, int max_out, float* output) {
    const int out_elem = 1 + max_out * DET_FLOATS;
    const int info_len = 4 + classes;
    for (int b = 0; b < batch; ++b) {
        float* out = output + (size_t)b * out_elem;
        memset(out, 0, sizeof(float) * out_elem);
        int count = 0;
        for (int l = 0; l < n_strides; ++l) {
            const int stride = strides[l];
            const int gh = net_h / stride, gw = net_w / stride; /* yololayer.cu:296-297 */
            const int total = gh * gw;
            const float* cur = inputs[l] + (size_t)b * total * info_len;
            for (int e = 0; e < total; ++e) {
                int class_id = 0;
                float max_p = 0.0f;
                for (int i = 4; i < 4 + classes; ++i) { /* yololayer.cu:195-201 */
                    float p = logist(cur[e + (size_t)i * total]);
                    if (p > max_p) {
                        max_p = p;
                        class_id = i - 4;
                    }
                }
                if (max_p < 0.1) /* double literal as in yololayer.cu:203 */
                    continue;
                if (count >= max_out) { /* slot >= maxoutobject: dropped (yololayer.cu:207) */
                    continue;
                }
                float* det = out + 1 + (size_t)count * DET_FLOATS;
                const int row = e / gw, col = e % gw;
                det[0] = (col + 0.5f - cur[e + 0 * (size_t)total]) * stride; /* yololayer.cu:217-220 */
                det[1] = (row + 0.5f - cur[e + 1 * (size_t)total]) * stride;
                det[2] = (col + 0.5f + cur[e + 2 * (size_t)total]) * stride;
                det[3] = (row + 0.5f + cur[e + 3 * (size_t)total]) * stride;
                det[4] = max_p;
                det[5] = (float)class_id;
                ++count;
            }
        }
        out[0] = (float)count;
    }
}

static float iou_xyxy(const float* l, const float* r) { /* postprocess.cpp:71-85 */
    float ib0 = l[0] > r[0] ? l[0] : r[0];
    float ib1 = l[2] < r[2] ? l[2] : r[2];
    float ib2 = l[1] > r[1] ? l[1] : r[1];
    float ib3 = l[3] < r[3] ? l[3] : r[3];
    if (ib2 > ib3 || ib0 > ib1)
        return 0.0f;
    float inter = (ib1 - ib0) * (ib3 - ib2);
    float uni = (l[2] - l[0]) * (l[3] - l[1]) + (r[2] - r[0]) * (r[3] - r[1]) - inter;
    return inter / uni;
}

typedef struct {
    const float* det;
    int idx;
} cand_t;

/* total order: class asc (std::map<float,...> iteration), conf desc, bbox[0] asc (cmp, :87-92), slot asc */
static int cand_less(const cand_t* a, const cand_t* b) {
    if (a->det[5] != b->det[5])
        return a->det[5] < b->det[5];
    if (a->det[4] != b->det[4])
        return a->det[4] > b->det[4];
    if (a->det[0] != b->det[0])
        return a->det[0] < b->det[0];
    return a->idx < b->idx;
}

static int cand_cmp_qsort(const void* pa, const void* pb) {
    const cand_t* a = (const cand_t*)pa;
    const cand_t* b = (const cand_t*)pb;
    if (cand_less(a, b))
        return -1;
    if (cand_less(b, a))
        return 1;
    return 0;
}

/* One image.  output: [1 + max_out*90].  keep_idx[k] = decode slot of the k-th kept detection,
 * in the reference's emission order (class asc, conf desc).  Returns number kept. */
int yolo_nms_ref(const float* output, int max_out, float conf_thresh, float nms_thresh, int* keep_idx,
                 float* keep_det /* [n][6] or NULL */) {
    int count = (int)output[0];
    if (count > max_out)
        count = max_out;
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (count > 0 ? count : 1));
    int n = 0;
    for (int i = 0; i < count; ++i) { /* postprocess.cpp:98-100 */
        const float* det = output + 1 + (size_t)DET_FLOATS * i;
        if (det[4] <= conf_thresh || isnan(det[4]))
            continue;
        c[n].det = det;
        c[n].idx = i;
        ++n;
    }
    qsort(c, n, sizeof(cand_t), cand_cmp_qsort); /* total order => deterministic */
    char* dead = (char*)calloc(n > 0 ? n : 1, 1);
    int kept = 0;
    for (int m = 0; m < n; ++m) { /* postprocess.cpp:109-120, per class bucket */
        if (dead[m])
            continue;
        keep_idx[kept] = c[m].idx;
        if (keep_det)
            memcpy(keep_det + 6 * (size_t)kept, c[m].det, 6 * sizeof(float));
        ++kept;
        for (int k = m + 1; k < n && c[k].det[5] == c[m].det[5]; ++k) {
            if (!dead[k] && iou_xyxy(c[m].det, c[k].det) > nms_thresh)
                dead[k] = 1;
        }
    }
    free(dead);
    free(c);
    return kept;
}

/* batch_nms, postprocess.cpp:123-129.  keep_idx: [batch][max_out], keep_cnt: [batch] */
void yolo_batch_nms_ref(const float* output, int batch, int max_out, float conf_thresh, float nms_thresh,
                        int* keep_idx, int* keep_cnt, float* keep_det /* [batch][max_out][6] or NULL */) {
    const int out_elem = 1 + max_out * DET_FLOATS;
    for (int b = 0; b < batch; ++b) {
        keep_cnt[b] = yolo_nms_ref(output + (size_t)b * out_elem, max_out, conf_thresh, nms_thresh,
                                   keep_idx + (size_t)b * max_out,
                                   keep_det ? keep_det + (size_t)b * max_out * 6 : NULL);
    }
}

/* ---- GPU post-processing mode "g" (yolov8/src/postprocess.cu:42-111; call site yolov8_det.cpp:105-112), restated
 * sequentially.  decode_kernel copies candidates with conf >= conf_thresh into 7-float records (x1,y1,x2,y2,conf,cls,keep=1);
 * every position takes a slot (atomicAdd before the confidence test), so slots of rejected positions stay zero and the
 * count is the input count.  Canonical slot order = input order (the reference's atomics make it arbitrary).
 * nms_kernel is NOT greedy: a box is dropped if ANY same-class box with higher confidence (equal confidence: higher
 * index) overlaps it by more than the threshold, whether or not that box survives itself.
 * out: [batch][1 + max_out*7] */
static float gpu_box_iou(const float* a, const float* b) {
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

void yolo_gpu_postprocess_ref(const float* output, int batch, int max_out, float conf_thresh, float nms_thresh, float* out) {
    const int in_elem = 1 + max_out * DET_FLOATS, out_elem = 1 + max_out * 7;
    for (int b = 0; b < batch; ++b) {
        const float* src = output + (size_t)b * in_elem;
        float* dst = out + (size_t)b * out_elem;
        memset(dst, 0, sizeof(float) * out_elem);
        int count = (int)src[0];
        if (count > max_out) count = max_out; /* decode-count clamp, as everywhere in this oracle */
        dst[0] = (float)count;
        for (int i = 0; i < count; ++i) {
            const float* it = src + 1 + (size_t)i * DET_FLOATS;
            if (it[4] < conf_thresh) continue; /* NaN passes "<", exactly as in the kernel */
            float* o = dst + 1 + (size_t)i * 7;
            o[0] = it[0]; o[1] = it[1]; o[2] = it[2]; o[3] = it[3]; o[4] = it[4]; o[5] = it[5]; o[6] = 1.0f;
        }
        for (int p = 0; p < count; ++p) {
            float* cur = dst + 1 + (size_t)p * 7;
            for (int i = 0; i < count; ++i) {
                const float* it = dst + 1 + (size_t)i * 7;
                if (i == p || cur[5] != it[5]) continue;
                if (it[4] >= cur[4]) {
                    if (it[4] == cur[4] && i < p) continue;
                    if (gpu_box_iou(cur, it) > nms_thresh) { cur[6] = 0.0f; break; }
                }
            }
        }
    }
}
