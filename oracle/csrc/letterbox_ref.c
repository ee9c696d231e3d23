/*
 * ORACLE — test infrastructure only.  Never linked into / called by the product path.
 * CPU restatement of the reference's GPU letterbox pre-processing, yolov8/src/preprocess.cu:7-117 (warpaffine_kernel and the
 * matrix set-up of cuda_preprocess; cv::invertAffineTransform restated from OpenCV's published CV_32F branch).
 * Pinned on the reference's own kernel (oracle/_ref/libref_yolov8_post.so, run on the MI355X) in tests/test_gpu_letterbox.py.
 * Build flags: -ffp-contract=off (oracle/Makefile): every float operation stays a single IEEE operation.
 */
#include <math.h>
#include <stdint.h>

void letterbox_matrix_ref(int src_w, int src_h, int dst_w, int dst_h, float* d2s) {
    const float a = dst_h / (float)src_h, b = dst_w / (float)src_w;
    const float scale = a < b ? a : b; /* std::min */
    float s2d[6];
    s2d[0] = scale;
    s2d[1] = 0;
    s2d[2] = -scale * src_w * 0.5 + dst_w * 0.5;
    s2d[3] = 0;
    s2d[4] = scale;
    s2d[5] = -scale * src_h * 0.5 + dst_h * 0.5;
    double D = (double)s2d[0] * s2d[4] - (double)s2d[1] * s2d[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = s2d[4] * D, A22 = s2d[0] * D, A12 = -s2d[1] * D, A21 = -s2d[3] * D;
    const double b1 = -A11 * s2d[2] - A12 * s2d[5];
    const double b2 = -A21 * s2d[2] - A22 * s2d[5];
    d2s[0] = (float)A11; d2s[1] = (float)A12; d2s[2] = (float)b1;
    d2s[3] = (float)A21; d2s[4] = (float)A22; d2s[5] = (float)b2;
}

/* src: uint8 HWC BGR [src_h][src_w][3]; dst: fp32 CHW RGB [3][dst_h][dst_w] */
void letterbox_ref(const uint8_t* src, int src_w, int src_h, float* dst, int dst_w, int dst_h) {
    float m[6];
    letterbox_matrix_ref(src_w, src_h, dst_w, dst_h, m);
    const int line = src_w * 3, area = dst_w * dst_h;
    const uint8_t cvl[3] = {128, 128, 128};
    for (int dy = 0; dy < dst_h; ++dy)
        for (int dx = 0; dx < dst_w; ++dx) {
            const float src_x = m[0] * dx + m[1] * dy + m[2] + 0.5f;
            const float src_y = m[3] * dx + m[4] * dy + m[5] + 0.5f;
            float c0, c1, c2;
            if (src_x <= -1 || src_x >= src_w || src_y <= -1 || src_y >= src_h) {
                c0 = c1 = c2 = 128;
            } else {
                const int y_low = (int)floorf(src_y), x_low = (int)floorf(src_x);
                const int y_high = y_low + 1, x_high = x_low + 1;
                const float ly = src_y - y_low, lx = src_x - x_low;
                const float hy = 1 - ly, hx = 1 - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const uint8_t *v1 = cvl, *v2 = cvl, *v3 = cvl, *v4 = cvl;
                if (y_low >= 0) {
                    if (x_low >= 0) v1 = src + (long)y_low * line + x_low * 3;
                    if (x_high < src_w) v2 = src + (long)y_low * line + x_high * 3;
                }
                if (y_high < src_h) {
                    if (x_low >= 0) v3 = src + (long)y_high * line + x_low * 3;
                    if (x_high < src_w) v4 = src + (long)y_high * line + x_high * 3;
                }
                c0 = w1 * v1[0] + w2 * v2[0] + w3 * v3[0] + w4 * v4[0];
                c1 = w1 * v1[1] + w2 * v2[1] + w3 * v3[1] + w4 * v4[1];
                c2 = w1 * v1[2] + w2 * v2[2] + w3 * v3[2] + w4 * v4[2];
            }
            const float t = c2;
            c2 = c0;
            c0 = t;
            dst[dy * dst_w + dx] = c0 / 255.0f;
            dst[area + dy * dst_w + dx] = c1 / 255.0f;
            dst[2 * area + dy * dst_w + dx] = c2 / 255.0f;
        }
}
