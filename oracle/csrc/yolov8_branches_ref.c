/*
 * ORACLE — test infrastructure only.  Never linked into / called by the product path.
 * CPU restatement of the seg / pose / obb branches of the YOLOv8 YoloLayer and of the oriented-box NMS:
 *   - decode : yolov8/plugin/yololayer.cu:178-279 (CalDetection with is_segmentation / is_pose / is_obb)
 *   - host NMS for oriented boxes: yolov8/src/postprocess.cpp:303-393 (convariance_matrix, probiou, nms_obb) with the
 *     reference's float/double promotions (std::pow(float, int) -> double, 12.0 / 1.0 literals, float cos/sin/sqrt/exp)
 *   - GPU mode for oriented boxes: yolov8/src/postprocess.cu:7-40 (decode_kernel_obb), :113-166 (box_probiou, nms_kernel_obb)
 * Pinned on the reference's own plugin / host functions / kernels in tests/test_gpu_yolo8_branches.py and test_ref_pinning.py.
 * Canonicalisations as in yolo_post_ref.c (slot order, clamped count, ties by slot).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DET 90

static float logistf(float x) { return 1.0f / (1.0f + expf(-x)); }

void yolo_decode_ex_ref(const float* const* inputs, int batch, int classes, int net_h, int net_w, const int* strides, int n_strides,
                        int max_out, int nk, float kpt_conf, int is_seg, int is_pose, int is_obb, float* output) {
    const int out_elem = 1 + max_out * DET;
    const int info_len = 4 + classes + (is_seg ? 32 : 0) + (is_pose ? nk * 3 : 0) + (is_obb ? 1 : 0);
    for (int b = 0; b < batch; ++b) {
        float* out = output + (size_t)b * out_elem;
        memset(out, 0, sizeof(float) * out_elem);
        int count = 0;
        for (int l = 0; l < n_strides; ++l) {
            const int stride = strides[l];
            const int gh = net_h / stride, gw = net_w / stride, total = gh * gw;
            const float* cur = inputs[l] + (size_t)b * total * info_len;
            for (int e = 0; e < total; ++e) {
                int class_id = 0;
                float max_p = 0.0f;
                for (int i = 4; i < 4 + classes; ++i) {
                    const float p = logistf(cur[e + (size_t)i * total]);
                    if (p > max_p) {
                        max_p = p;
                        class_id = i - 4;
                    }
                }
                if (max_p < 0.1) continue;
                if (count >= max_out) continue;
                float* det = out + 1 + (size_t)count * DET;
                const int row = e / gw, col = e % gw;
                det[4] = max_p;
                det[5] = (float)class_id;
                det[0] = (col + 0.5f - cur[e + 0 * (size_t)total]) * stride;
                det[1] = (row + 0.5f - cur[e + 1 * (size_t)total]) * stride;
                det[2] = (col + 0.5f + cur[e + 2 * (size_t)total]) * stride;
                det[3] = (row + 0.5f + cur[e + 3 * (size_t)total]) * stride;
                if (is_seg)
                    for (int k = 0; k < 32; ++k)
                        det[6 + k] = cur[e + (size_t)(4 + classes + (is_pose ? nk * 3 : 0) + (is_obb ? 1 : 0) + k) * total];
                if (is_pose)
                    for (int kpt = 0; kpt < nk; ++kpt) {
                        const size_t base = (size_t)(4 + classes + (is_seg ? 32 : 0) + (is_obb ? 1 : 0) + kpt * 3) * total;
                        const float kc = logistf(cur[e + base + 2 * (size_t)total]);
                        const float kx = (cur[e + base] * 2.0 + col) * stride;
                        const float ky = (cur[e + base + (size_t)total] * 2.0 + row) * stride;
                        const int inside = kx >= det[0] && kx <= det[2] && ky >= det[1] && ky <= det[3];
                        float* o = det + 38 + kpt * 3;
                        if (kc < kpt_conf || !inside) {
                            o[0] = o[1] = o[2] = -1;
                        } else {
                            o[0] = kx;
                            o[1] = ky;
                            o[2] = kc;
                        }
                    }
                if (is_obb) {
                    const double pi = M_PI;
                    const float ain = cur[e + (size_t)(4 + classes + (is_seg ? 32 : 0) + (is_pose ? nk * 3 : 0)) * total];
                    const double angle = (logistf(ain) - 0.25f) * pi;
                    const double cos1 = cos(angle), sin1 = sin(angle);
                    const float xf = (cur[e + 2 * (size_t)total] - cur[e + 0 * (size_t)total]) / 2;
                    const float yf = (cur[e + 3 * (size_t)total] - cur[e + 1 * (size_t)total]) / 2;
                    const double x = xf * cos1 - yf * sin1;
                    const double y = xf * sin1 + yf * cos1;
                    const float cx = (col + 0.5f + x) * stride;
                    const float cy = (row + 0.5f + y) * stride;
                    const float w1 = (cur[e + 0 * (size_t)total] + cur[e + 2 * (size_t)total]) * stride;
                    const float h1 = (cur[e + 1 * (size_t)total] + cur[e + 3 * (size_t)total]) * stride;
                    det[0] = cx;
                    det[1] = cy;
                    det[2] = w1;
                    det[3] = h1;
                    det[DET - 1] = angle;
                }
                ++count;
            }
        }
        out[0] = (float)count;
    }
}

/* postprocess.cpp:303-322 */
static void cov_host(const float* det, float* a_val, float* b_val, float* c_val) {
    const float w = det[2], h = det[3];
    const float a = w * w / 12.0, b = h * h / 12.0, c = det[DET - 1];
    const float cos_r = cosf(c), sin_r = sinf(c);
    const float cos_r2 = cos_r * cos_r, sin_r2 = sin_r * sin_r;
    *a_val = a * cos_r2 + b * sin_r2;
    *b_val = a * sin_r2 + b * cos_r2;
    *c_val = (a - b) * cos_r * sin_r;
}
/* postprocess.cpp:324-355; std::pow(float, int) is the double pow */
static float probiou_host(const float* r1, const float* r2) {
    const float eps = 1e-7f;
    float a1, b1, c1, a2, b2, c2;
    cov_host(r1, &a1, &b1, &c1);
    cov_host(r2, &a2, &b2, &c2);
    const float x1 = r1[0], y1 = r1[1], x2 = r2[0], y2 = r2[1];
    const float t1 = ((a1 + a2) * pow(y1 - y2, 2) + (b1 + b2) * pow(x1 - x2, 2)) / ((a1 + a2) * (b1 + b2) - pow(c1 + c2, 2) + eps);
    const float t2 = ((c1 + c2) * (x2 - x1) * (y1 - y2)) / ((a1 + a2) * (b1 + b2) - pow(c1 + c2, 2) + eps);
    const float m1 = a1 * b1 - c1 * c1, m2 = a2 * b2 - c2 * c2;
    const float t3 = log(((a1 + a2) * (b1 + b2) - pow(c1 + c2, 2)) / (4 * sqrtf(m1 > 0.0f ? m1 : 0.0f) * sqrtf(m2 > 0.0f ? m2 : 0.0f) + eps) + eps);
    float bd = 0.25f * t1 + 0.5f * t2 + 0.5f * t3;
    bd = bd < 100.0f ? bd : 100.0f;
    bd = bd > eps ? bd : eps;
    const float hd = sqrt(1.0 - expf(-bd) + eps);
    return 1 - hd;
}

typedef struct {
    float cls, conf, x0;
    int slot;
} co_t;
static int co_cmp(const void* pa, const void* pb) {
    const co_t *a = (const co_t*)pa, *b = (const co_t*)pb;
    if (a->cls != b->cls) return a->cls < b->cls ? -1 : 1;
    if (a->conf != b->conf) return a->conf > b->conf ? -1 : 1; /* cmp(): conf desc, then bbox[0] asc */
    if (a->x0 != b->x0) return a->x0 < b->x0 ? -1 : 1;
    return a->slot < b->slot ? -1 : (a->slot > b->slot ? 1 : 0);
}
/* nms_obb for one image; keep_det [n][7] = cx, cy, w, h, conf, cls, angle */
int yolo_nms_obb_ref(const float* output, int max_out, float conf_thresh, float nms_thresh, int* keep_idx, float* keep_det) {
    int count = (int)output[0];
    if (count > max_out) count = max_out;
    co_t* c = (co_t*)malloc(sizeof(co_t) * (count > 0 ? count : 1));
    int n = 0;
    for (int i = 0; i < count; ++i) {
        const float* det = output + 1 + (size_t)i * DET;
        if (det[4] <= conf_thresh) continue;
        c[n].cls = det[5]; c[n].conf = det[4]; c[n].x0 = det[0]; c[n].slot = i;
        ++n;
    }
    qsort(c, n, sizeof(co_t), co_cmp);
    char* dead = (char*)calloc(n > 0 ? n : 1, 1);
    int kept = 0;
    for (int m = 0; m < n; ++m) {
        if (dead[m]) continue;
        const float* item = output + 1 + (size_t)c[m].slot * DET;
        keep_idx[kept] = c[m].slot;
        if (keep_det) {
            memcpy(keep_det + (size_t)kept * 7, item, 6 * sizeof(float));
            keep_det[(size_t)kept * 7 + 6] = item[DET - 1];
        }
        ++kept;
        for (int q = m + 1; q < n && c[q].cls == c[m].cls; ++q)
            if (!dead[q] && probiou_host(item, output + 1 + (size_t)c[q].slot * DET) >= nms_thresh) dead[q] = 1;
    }
    free(c);
    free(dead);
    return kept;
}

/* GPU mode (postprocess.cu): float math */
static void cov_g(float w, float h, float r, float* a, float* b, float* c) {
    const float a_val = w * w / 12.0f, b_val = h * h / 12.0f;
    const float cos_r = cosf(r), sin_r = sinf(r);
    *a = a_val * cos_r * cos_r + b_val * sin_r * sin_r;
    *b = a_val * sin_r * sin_r + b_val * cos_r * cos_r;
    *c = (a_val - b_val) * sin_r * cos_r;
}
static float probiou_g(const float* p, const float* q) {
    const float eps = 1e-7f;
    float a1, b1, c1, a2, b2, c2;
    cov_g(p[2], p[3], p[7], &a1, &b1, &c1);
    cov_g(q[2], q[3], q[7], &a2, &b2, &c2);
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
/* decode_kernel_obb + nms_kernel_obb in input order: out [batch][1 + max_out*8] */
void yolo_gpu_postprocess_obb_ref(const float* output, int batch, int max_out, float conf_thresh, float nms_thresh, float* out) {
    for (int b = 0; b < batch; ++b) {
        const float* img = output + (size_t)b * (1 + (size_t)max_out * DET);
        float* dst = out + (size_t)b * (1 + (size_t)max_out * 8);
        memset(dst, 0, sizeof(float) * (1 + (size_t)max_out * 8));
        int count = (int)img[0];
        if (count > max_out) count = max_out;
        dst[0] = (float)count;
        float* rec = dst + 1;
        for (int p = 0; p < count; ++p) {
            const float* it = img + 1 + (size_t)p * DET;
            if (it[4] < conf_thresh) continue;
            memcpy(rec + (size_t)p * 8, it, 6 * sizeof(float));
            rec[(size_t)p * 8 + 6] = 1.0f;
            rec[(size_t)p * 8 + 7] = it[DET - 1];
        }
        float* keep = (float*)malloc(sizeof(float) * (count > 0 ? count : 1));
        for (int p = 0; p < count; ++p) {
            const float* cur = rec + (size_t)p * 8;
            keep[p] = cur[6];
            for (int i = 0; i < count; ++i) {
                const float* it = rec + (size_t)i * 8;
                if (i == p || cur[5] != it[5]) continue;
                if (it[4] >= cur[4]) {
                    if (it[4] == cur[4] && i < p) continue;
                    if (probiou_g(cur, it) > nms_thresh) {
                        keep[p] = 0.0f;
                        break;
                    }
                }
            }
        }
        for (int p = 0; p < count; ++p) rec[(size_t)p * 8 + 6] = keep[p];
        free(keep);
    }
}
