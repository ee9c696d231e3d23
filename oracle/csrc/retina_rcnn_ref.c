/*
 * ORACLE — test infrastructure only.  Never linked into / called by the product path.
 *
 * Sequential plain-C restatements of the RetinaFace and R-CNN (detectron2-style) plugins of
 * wang-xinyu/tensorrtx.  Parity status: "parity unpinned" (the reference has no golden vectors; the .cu
 * files need nvcc + TensorRT and cannot be built here).  Each function cites the lines it follows.
 *
 * Canonicalisations (reference behaviour is racy / undefined there):
 *   - RetinaFace decode slots are handed out by atomicAdd (decode.cu:132); canonical order here is
 *     (level, cell, k) ascending.
 *   - cub::DeviceRadixSort::SortPairsDescending is stable: equal keys keep ascending index order; that is what
 *     "sort" means below (-0.0 is treated as equal to +0.0).
 *   - rpn_nms_kernel / batched_nms_kernel are launched multi-block with a block-level barrier
 *     (RpnNms.cu:108, BatchedNms.cu:142): the intended semantics, exact sequential greedy, is restated.
 *   - std::sort in the RetinaFace host nms (common.hpp:118) is unstable: ties broken by slot index.
 *
 * Build: gcc -O2 -ffp-contract=off (every float op one IEEE operation, as the HIP plugin kernels).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ RetinaFace */
#define RF_DET 15 /* decodeplugin::Detection: bbox[4], class_confidence, landmark[10] (decode.h:11-15) */

/* inputs[l]: [batch][32][h*w] = bbox(2x4) | cls(2x2) | lmk(2x10) planes (retina_r50.cpp:197-202);
 * output: [batch][1 + anchors*15].  decode.cu:110-165 (CalDetection) and :167-191 (forwardGpu). */
void retina_decode_ref(const float* const* inputs, int batch, int net_h, int net_w, float* output) {
    int out_elem = 1;
    for (int s = 8; s <= 32; s *= 2) out_elem += (net_h / s) * (net_w / s) * 2 * RF_DET;
    for (int b = 0; b < batch; ++b) {
        float* out = output + (size_t)b * out_elem;
        memset(out, 0, sizeof(float) * out_elem);
        int count = 0;
        int step = 8, anchor = 16;
        for (int l = 0; l < 3; ++l, step *= 2, anchor *= 4) {
            const int h = net_h / step, w = net_w / step;
            const int total = h * w;
            const float* cur = inputs[l] + (size_t)b * 32 * total;
            const float* bbox_reg = cur;
            const float* cls_reg = cur + 2 * 4 * total;
            const float* lmk_reg = cur + 2 * 4 * total + 2 * 2 * total;
            for (int idx = 0; idx < total; ++idx) {
                const int y = idx / w, x = idx % w;
                for (int k = 0; k < 2; ++k) {
                    float conf1 = cls_reg[idx + k * total * 2];
                    float conf2 = cls_reg[idx + k * total * 2 + total];
                    conf2 = expf(conf2) / (expf(conf1) + expf(conf2));
                    if (conf2 <= 0.02) continue; /* double literal, decode.cu:129 */
                    float* det = out + 1 + (size_t)count * RF_DET;
                    ++count;
                    float prior[4];
                    prior[0] = ((float)x + 0.5) / w; /* double arithmetic, rounded on store */
                    prior[1] = ((float)y + 0.5) / h;
                    prior[2] = (float)anchor * (k + 1) / net_w;
                    prior[3] = (float)anchor * (k + 1) / net_h;
                    float bb0 = prior[0] + bbox_reg[idx + k * total * 4] * 0.1 * prior[2];
                    float bb1 = prior[1] + bbox_reg[idx + k * total * 4 + total] * 0.1 * prior[3];
                    float bb2 = prior[2] * expf(bbox_reg[idx + k * total * 4 + total * 2] * 0.2);
                    float bb3 = prior[3] * expf(bbox_reg[idx + k * total * 4 + total * 3] * 0.2);
                    bb0 -= bb2 / 2;
                    bb1 -= bb3 / 2;
                    bb2 += bb0;
                    bb3 += bb1;
                    bb0 *= net_w;
                    bb1 *= net_h;
                    bb2 *= net_w;
                    bb3 *= net_h;
                    det[0] = bb0; det[1] = bb1; det[2] = bb2; det[3] = bb3;
                    det[4] = conf2;
                    for (int i = 0; i < 10; i += 2) {
                        float lx = prior[0] + lmk_reg[idx + k * total * 10 + total * i] * 0.1 * prior[2];
                        float ly = prior[1] + lmk_reg[idx + k * total * 10 + total * (i + 1)] * 0.1 * prior[3];
                        lx *= net_w;
                        ly *= net_h;
                        det[5 + i] = lx;
                        det[5 + i + 1] = ly;
                    }
                }
            }
        }
        out[0] = (float)count;
    }
}

static float rf_iou(const float* l, const float* r) { /* common.hpp:91-104 */
    float i0 = l[0] > r[0] ? l[0] : r[0];
    float i1 = l[2] < r[2] ? l[2] : r[2];
    float i2 = l[1] > r[1] ? l[1] : r[1];
    float i3 = l[3] < r[3] ? l[3] : r[3];
    if (i2 > i3 || i0 > i1) return 0.0f;
    float inter = (i1 - i0) * (i3 - i2);
    return inter / ((l[2] - l[0]) * (l[3] - l[1]) + (r[2] - r[0]) * (r[3] - r[1]) - inter + 0.000001f);
}

typedef struct { float key; int idx; } kv_t;
static int kv_desc(const void* a, const void* b) {
    const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
    if (x->key != y->key) return x->key > y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* common.hpp:110-130 for one image; returns number kept; keep_idx = decode slots in emission order */
int retina_nms_ref(const float* output, double conf_thresh, float nms_thresh, int max_keep, int* keep_idx) {
    const int count = (int)output[0];
    kv_t* c = (kv_t*)malloc(sizeof(kv_t) * (count > 0 ? count : 1));
    int n = 0;
    for (int i = 0; i < count; ++i) {
        if ((double)output[RF_DET * i + 1 + 4] <= conf_thresh) continue; /* "<= 0.1" with a double literal (common.hpp:113) */
        c[n].key = output[RF_DET * i + 1 + 4];
        c[n].idx = i;
        ++n;
    }
    qsort(c, n, sizeof(kv_t), kv_desc);
    char* dead = (char*)calloc(n > 0 ? n : 1, 1);
    int kept = 0;
    for (int m = 0; m < n; ++m) {
        if (dead[m]) continue;
        if (kept < max_keep) keep_idx[kept] = c[m].idx;
        ++kept;
        const float* mb = output + 1 + (size_t)RF_DET * c[m].idx;
        for (int k = m + 1; k < n; ++k)
            if (!dead[k] && rf_iou(mb, output + 1 + (size_t)RF_DET * c[k].idx) > nms_thresh) dead[k] = 1;
    }
    free(dead);
    free(c);
    return kept < max_keep ? kept : max_keep;
}

/* ---------------------------------------------------------------------------------------------- R-CNN */
/* stable descending argsort of n floats (cub::DeviceRadixSort::SortPairsDescending) */
static void argsort_desc(const float* keys, int n, int* order) {
    kv_t* c = (kv_t*)malloc(sizeof(kv_t) * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { c[i].key = keys[i] + 0.0f; c[i].idx = i; }
    qsort(c, n, sizeof(kv_t), kv_desc);
    for (int i = 0; i < n; ++i) order[i] = c[i].idx;
    free(c);
}

/* RpnDecode.cu:27-143.  scores [A*h*w], deltas [A*4*h*w] per image; anchors [A*4]. */
void rpn_decode_ref(int batch, const float* scores, const float* deltas, int height, int width, int image_height,
                    int image_width, float stride, const float* anchors, int num_anchors, int top_n, float* out_scores,
                    float* out_boxes) {
    const int scores_size = num_anchors * height * width;
    int* order = (int*)malloc(sizeof(int) * scores_size);
    for (int b = 0; b < batch; ++b) {
        const float* in_scores = scores + (size_t)b * scores_size;
        const float* in_boxes = deltas + (size_t)b * scores_size * 4;
        float* os = out_scores + (size_t)b * top_n;
        float* ob = out_boxes + (size_t)b * top_n * 4;
        int num = scores_size;
        if (num > top_n) {
            argsort_desc(in_scores, scores_size, order);
            num = top_n;
        } else {
            for (int i = 0; i < scores_size; ++i) order[i] = i;
        }
        for (int d = 0; d < num; ++d) {
            const int i = order[d];
            const int x = i % width, y = (i / width) % height, a = (i / height / width) % num_anchors;
            float bx = in_boxes[((a * 4 + 0) * height + y) * width + x];
            float by = in_boxes[((a * 4 + 1) * height + y) * width + x];
            float bz = in_boxes[((a * 4 + 2) * height + y) * width + x];
            float bw = in_boxes[((a * 4 + 3) * height + y) * width + x];
            const float fx = x * stride, fy = y * stride;
            const float* dd = anchors + 4 * a;
            const float x1 = fx + dd[0], y1 = fy + dd[1], x2 = fx + dd[2], y2 = fy + dd[3];
            const float w = x2 - x1, h = y2 - y1;
            const float pcx = bx * w + x1 + 0.5f * w;
            const float pcy = by * h + y1 + 0.5f * h;
            const float pw = expf(bz) * w;
            const float ph = expf(bw) * h;
            float r0 = pcx - 0.5f * pw; r0 = r0 > 0.0f ? r0 : 0.0f;
            float r1 = pcy - 0.5f * ph; r1 = r1 > 0.0f ? r1 : 0.0f;
            float r2 = pcx + 0.5f * pw; r2 = r2 < (float)image_width ? r2 : (float)image_width;
            float r3 = pcy + 0.5f * ph; r3 = r3 < (float)image_height ? r3 : (float)image_height;
            ob[4 * d + 0] = r0; ob[4 * d + 1] = r1; ob[4 * d + 2] = r2; ob[4 * d + 3] = r3;
            os[d] = (r2 - r0 <= 0.0f || r3 - r1 <= 0.0f) ? -FLT_MAX : in_scores[i];
        }
        for (int d = num; d < top_n; ++d) {
            os[d] = -FLT_MAX;
            ob[4 * d + 0] = ob[4 * d + 1] = ob[4 * d + 2] = ob[4 * d + 3] = 0.0f;
        }
    }
    free(order);
}

static float plain_iou(const float* i, const float* m) { /* RpnNms.cu:38-48, BatchedNms.cu:42-52 */
    float x1 = i[0] > m[0] ? i[0] : m[0];
    float y1 = i[1] > m[1] ? i[1] : m[1];
    float x2 = i[2] < m[2] ? i[2] : m[2];
    float y2 = i[3] < m[3] ? i[3] : m[3];
    float w = x2 - x1; w = w > 0.0f ? w : 0.0f;
    float h = y2 - y1; h = h > 0.0f ? h : 0.0f;
    float iarea = (i[2] - i[0]) * (i[3] - i[1]);
    float marea = (m[2] - m[0]) * (m[3] - m[1]);
    float inter = w * h;
    return inter / (iarea + marea - inter);
}

/* RpnNms.cu:59-121: sort desc, exact greedy suppression (score -> -FLT_MAX), stable re-sort, first post boxes */
void rpn_nms_ref(int batch, const float* scores, const float* boxes, int pre, int post, float thresh, float* out_boxes) {
    int* ord = (int*)malloc(sizeof(int) * pre);
    int* ord2 = (int*)malloc(sizeof(int) * pre);
    float* s = (float*)malloc(sizeof(float) * pre);
    for (int b = 0; b < batch; ++b) {
        const float* is = scores + (size_t)b * pre;
        const float* ib = boxes + (size_t)b * pre * 4;
        argsort_desc(is, pre, ord);
        for (int i = 0; i < pre; ++i) s[i] = is[ord[i]];
        for (int m = 0; m < pre; ++m) {
            if (!(s[m] > -FLT_MAX)) continue;
            for (int i = m + 1; i < pre; ++i)
                if (plain_iou(ib + 4 * ord[i], ib + 4 * ord[m]) > thresh) s[i] = -FLT_MAX;
        }
        argsort_desc(s, pre, ord2); /* positions in the first sorted order */
        const int n = post < pre ? post : pre;
        for (int d = 0; d < n; ++d) memcpy(out_boxes + ((size_t)b * post + d) * 4, ib + 4 * ord[ord2[d]], 16);
        for (int d = n; d < post; ++d) memset(out_boxes + ((size_t)b * post + d) * 4, 0, 16);
    }
    free(ord); free(ord2); free(s);
}

/* RoiAlign.cu:29-80 */
static float bilinear(const float* data, int height, int width, float y, float x) {
    if (y < -1.0 || y > height || x < -1.0 || x > width) return 0;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    float ly = y - y_low, lx = x - x_low;
    float hy = 1. - ly, hx = 1. - lx;
    float v1 = data[y_low * width + x_low], v2 = data[y_low * width + x_high];
    float v3 = data[y_high * width + x_low], v4 = data[y_high * width + x_high];
    float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* RoiAlign.cu:83-182.  boxes [B][P][4], features [B][C][fh][fw] -> out [B][P][C][res][res] */
void roi_align_ref(int batch, const float* boxes, const float* features, int res, float spatial_scale, int sampling_ratio,
                   int num_proposals, int channels, int fh, int fw, float* out) {
    for (int b = 0; b < batch; ++b) {
        const float* rois = boxes + (size_t)b * num_proposals * 4;
        const float* feat = features + (size_t)b * channels * fh * fw;
        float* top = out + (size_t)b * num_proposals * channels * res * res;
        for (int n = 0; n < num_proposals; ++n) {
            const float* r = rois + 4 * n;
            const float roi_offset = 0.5f;
            const float sw = r[0] * spatial_scale - roi_offset, sh = r[1] * spatial_scale - roi_offset;
            const float ew = r[2] * spatial_scale - roi_offset, eh = r[3] * spatial_scale - roi_offset;
            const float roi_w = ew - sw, roi_h = eh - sh;
            const float bin_h = roi_h / (float)res, bin_w = roi_w / (float)res;
            const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / res);
            const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / res);
            const float count = gh * gw;
            for (int c = 0; c < channels; ++c) {
                const float* plane = feat + (size_t)c * fh * fw;
                for (int ph = 0; ph < res; ++ph)
                    for (int pw = 0; pw < res; ++pw) {
                        float acc = 0.f;
                        for (int iy = 0; iy < gh; ++iy) {
                            const float y = sh + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
                            for (int ix = 0; ix < gw; ++ix) {
                                const float x = sw + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
                                acc += bilinear(plane, fh, fw, y, x);
                            }
                        }
                        acc /= count;
                        top[(((size_t)n * channels + c) * res + ph) * res + pw] = acc;
                    }
            }
        }
    }
}

/* PredictorDecode.cu:24-110.  scores [B][N*C], deltas [B][N*C][4], proposals [B][N][4] */
void predictor_decode_ref(int batch, const float* scores, const float* deltas, const float* proposals, int num_boxes,
                          int num_classes, int image_height, int image_width, const float* wts, float* out_scores,
                          float* out_boxes, float* out_classes) {
    const int scores_size = num_boxes * num_classes;
    int* order = (int*)malloc(sizeof(int) * scores_size);
    (void)image_height; /* reference clips y2 with image_width (PredictorDecode.cu:99) — kept */
    for (int b = 0; b < batch; ++b) {
        const float* is = scores + (size_t)b * scores_size;
        const float* id = deltas + (size_t)b * scores_size * 4;
        const float* ip = proposals + (size_t)b * num_boxes * 4;
        argsort_desc(is, scores_size, order);
        for (int d = 0; d < num_boxes; ++d) {
            const int i = order[d];
            const int cls = i % num_classes, n = i / num_classes;
            const float* dl = id + 4 * (size_t)i;
            const float* bx = ip + 4 * (size_t)n;
            const float w = bx[2] - bx[0], h = bx[3] - bx[1];
            const float pcx = (dl[0] / wts[0]) * w + bx[0] + 0.5f * w;
            const float pcy = (dl[1] / wts[1]) * h + bx[1] + 0.5f * h;
            const float pw = expf(dl[2] / wts[2]) * w;
            const float ph = expf(dl[3] / wts[3]) * h;
            float r0 = pcx - 0.5f * pw; r0 = r0 > 0.0f ? r0 : 0.0f;
            float r1 = pcy - 0.5f * ph; r1 = r1 > 0.0f ? r1 : 0.0f;
            float r2 = pcx + 0.5f * pw; r2 = r2 < (float)image_width ? r2 : (float)image_width;
            float r3 = pcy + 0.5f * ph; r3 = r3 < (float)image_width ? r3 : (float)image_width;
            float* ob = out_boxes + ((size_t)b * num_boxes + d) * 4;
            ob[0] = r0; ob[1] = r1; ob[2] = r2; ob[3] = r3;
            out_scores[(size_t)b * num_boxes + d] = (r2 - r0 <= 0.0f || r3 - r1 <= 0.0f) ? 0.0f : is[i];
            out_classes[(size_t)b * num_boxes + d] = (float)cls;
        }
    }
    free(order);
}

/* BatchedNms.cu:28-162.  method 0 hard / 1 soft-linear / 2 soft-gaussian */
void batched_nms_ref(int method, int batch, const float* scores, const float* boxes, const float* classes, int count,
                     int dets, float thresh, float* out_scores, float* out_boxes, float* out_classes) {
    int* ord = (int*)malloc(sizeof(int) * count);
    int* ord2 = (int*)malloc(sizeof(int) * count);
    float* s = (float*)malloc(sizeof(float) * count);
    for (int b = 0; b < batch; ++b) {
        const float* is = scores + (size_t)b * count;
        const float* ib = boxes + (size_t)b * count * 4;
        const float* ic = classes + (size_t)b * count;
        argsort_desc(is, count, ord);
        for (int i = 0; i < count; ++i) s[i] = is[ord[i]];
        for (int m = 0; m < count; ++m) {
            if (!(s[m] > 0.0f)) continue;
            for (int i = m + 1; i < count; ++i) {
                if ((int)ic[ord[m]] != (int)ic[ord[i]]) continue;
                const float ov = plain_iou(ib + 4 * ord[i], ib + 4 * ord[m]);
                if (!(ov > thresh)) continue;
                if (method == 1) s[i] = (1 - ov) * s[i];
                else if (method == 2) { const float sigma = 0.5; s[i] = expf(-(ov * ov) / sigma) * s[i]; }
                else s[i] = 0.0f;
            }
        }
        argsort_desc(s, count, ord2);
        const int n = dets < count ? dets : count;
        for (int d = 0; d < n; ++d) {
            out_scores[(size_t)b * dets + d] = s[ord2[d]];
            memcpy(out_boxes + ((size_t)b * dets + d) * 4, ib + 4 * ord[ord2[d]], 16);
            out_classes[(size_t)b * dets + d] = ic[ord[ord2[d]]];
        }
        for (int d = n; d < dets; ++d) {
            out_scores[(size_t)b * dets + d] = 0.0f;
            memset(out_boxes + ((size_t)b * dets + d) * 4, 0, 16);
            out_classes[(size_t)b * dets + d] = 0.0f;
        }
    }
    free(ord); free(ord2); free(s);
}
