/*
 * ORACLE — test infrastructure only.  Never linked into / called by the product path.
 *
 * CPU restatement (plain C, sequential semantics) of the anchor-based YOLO post-processing of wang-xinyu/tensorrtx:
 *   - decode : yolov5/plugin/yololayer.cu:161-227 (Logist, CalDetection, forwardGpu)
 *   - Detection record: yolov5/src/types.h:11-16 (bbox[4] centre format, conf, class_id, mask[32] = 38 floats)
 *   - NMS    : yolov5/src/postprocess.cpp:30-80 (iou on centre-format boxes, cmp by conf only, nms, batch_nms)
 * Pinned on the reference's own code: tests/test_ref_pinning.py runs the reference plugin (oracle/_ref/libref_yolov5_plugin.so,
 * on the MI355X) and the reference host nms (libref_host.so) against these functions.
 *
 * Canonicalisations (as for YOLOv8): slot order = (level, cell, anchor) ascending instead of the atomicAdd race
 * (yololayer.cu:193); stored count clamped to max_out; ties of (class, conf) in nms() broken by slot (std::sort is unstable).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DET5 38
#define NUM_ANCHOR 3 /* kNumAnchor, yolov5/src/config.h:35 */

static float logist5(float x) { return 1.0f / (1.0f + expf(-x)); } /* yololayer.cu:159 */

/* inputs[l]: [batch][3 * info_len][gh*gw]; anchors: [n_levels][6]; output [batch][1 + max_out*38] */
void yolov5_decode_ref(const float* const* inputs, int n_levels, int batch, int classes, int net_h, int net_w, const int* grid_w,
                       const int* grid_h, const float* anchors, int max_out, int is_seg, float* output) {
    const int out_elem = 1 + max_out * DET5;
    const int info_len = 5 + classes + (is_seg ? 32 : 0);
    for (int b = 0; b < batch; ++b) {
        float* out = output + (size_t)b * out_elem;
        memset(out, 0, sizeof(float) * out_elem);
        int count = 0;
        for (int l = 0; l < n_levels; ++l) {
            const int yw = grid_w[l], yh = grid_h[l], total = yw * yh;
            const float* cur = inputs[l] + (size_t)b * info_len * total * NUM_ANCHOR;
            for (int idx = 0; idx < total; ++idx)
                for (int k = 0; k < NUM_ANCHOR; ++k) {
                    const float* a = cur + idx + (size_t)k * info_len * total;
                    const float box_prob = logist5(a[(size_t)4 * total]);
                    if (box_prob < 0.1f) continue; /* kIgnoreThresh */
                    int class_id = 0;
                    float max_cls_prob = 0.0f;
                    for (int i = 5; i < 5 + classes; ++i) {
                        const float p = logist5(a[(size_t)i * total]);
                        if (p > max_cls_prob) {
                            max_cls_prob = p;
                            class_id = i - 5;
                        }
                    }
                    if (count < max_out) {
                        float* det = out + 1 + (size_t)count * DET5;
                        const int row = idx / yw, col = idx % yw;
                        det[0] = (col - 0.5f + 2.0f * logist5(a[0])) * net_w / yw;
                        det[1] = (row - 0.5f + 2.0f * logist5(a[(size_t)total])) * net_h / yh;
                        det[2] = 2.0f * logist5(a[(size_t)2 * total]);
                        det[2] = det[2] * det[2] * anchors[l * 6 + 2 * k];
                        det[3] = 2.0f * logist5(a[(size_t)3 * total]);
                        det[3] = det[3] * det[3] * anchors[l * 6 + 2 * k + 1];
                        det[4] = box_prob * max_cls_prob;
                        det[5] = (float)class_id;
                        for (int i = 0; is_seg && i < 32; ++i) det[6 + i] = a[(size_t)(i + 5 + classes) * total];
                    }
                    ++count;
                }
        }
        out[0] = (float)(count < max_out ? count : max_out);
    }
}

static float iou_cxcywh(const float* l, const float* r) { /* postprocess.cpp:30-44 */
    float ib0 = l[0] - l[2] / 2.f, t = r[0] - r[2] / 2.f;
    if (ib0 < t) ib0 = t;
    float ib1 = l[0] + l[2] / 2.f;
    t = r[0] + r[2] / 2.f;
    if (t < ib1) ib1 = t;
    float ib2 = l[1] - l[3] / 2.f;
    t = r[1] - r[3] / 2.f;
    if (ib2 < t) ib2 = t;
    float ib3 = l[1] + l[3] / 2.f;
    t = r[1] + r[3] / 2.f;
    if (t < ib3) ib3 = t;
    if (ib2 > ib3 || ib0 > ib1) return 0.0f;
    const float inter = (ib1 - ib0) * (ib3 - ib2);
    return inter / (l[2] * l[3] + r[2] * r[3] - inter);
}

typedef struct {
    float cls, conf;
    int slot;
} c5_t;

static int c5_cmp(const void* pa, const void* pb) {
    const c5_t *a = (const c5_t*)pa, *b = (const c5_t*)pb;
    if (a->cls != b->cls) return a->cls < b->cls ? -1 : 1;    /* std::map<float, ...>: class ascending */
    if (a->conf != b->conf) return a->conf > b->conf ? -1 : 1; /* cmp(): conf descending */
    return a->slot < b->slot ? -1 : (a->slot > b->slot ? 1 : 0);
}

/* one image; keep_idx: kept slot indices in emission order (class asc, conf desc); returns the count */
int yolov5_nms_ref(const float* output, int max_out, float conf_thresh, float nms_thresh, int* keep_idx, float* keep_det) {
    int count = (int)output[0];
    if (count > max_out) count = max_out; /* "i < output[0] && i < kMaxNumOutputBbox" */
    c5_t* c = (c5_t*)malloc(sizeof(c5_t) * (count > 0 ? count : 1));
    int n = 0;
    for (int i = 0; i < count; ++i) {
        const float* det = output + 1 + (size_t)i * DET5;
        if (det[4] <= conf_thresh) continue;
        c[n].cls = det[5];
        c[n].conf = det[4];
        c[n].slot = i;
        ++n;
    }
    qsort(c, n, sizeof(c5_t), c5_cmp);
    char* dead = (char*)calloc(n > 0 ? n : 1, 1);
    int kept = 0;
    for (int m = 0; m < n; ++m) {
        if (dead[m]) continue;
        const float* item = output + 1 + (size_t)c[m].slot * DET5;
        keep_idx[kept] = c[m].slot;
        if (keep_det) memcpy(keep_det + (size_t)kept * 6, item, 6 * sizeof(float));
        ++kept;
        for (int q = m + 1; q < n && c[q].cls == c[m].cls; ++q)
            if (!dead[q] && iou_cxcywh(item, output + 1 + (size_t)c[q].slot * DET5) > nms_thresh) dead[q] = 1;
    }
    free(c);
    free(dead);
    return kept;
}
