// Letterbox pre-processing for gfx950 (MI355X): uint8 HWC BGR image -> fp32 CHW RGB / 255, bilinear warp-affine with the
// grey (128) border - the reference's cuda_preprocess / warpaffine_kernel (yolov8/src/preprocess.cu:7-127).
//
// Same arithmetic per output pixel (this directory is built with -ffp-contract=off): source position from the inverse affine
// map + 0.5, out-of-range test, the four-neighbour weights, channel swap, / 255.  What changes is the orchestration:
//   * the whole batch is ONE launch (grid.y = image): per-image source pointer, size and inverse map travel in the kernel
//     argument block, where the reference launches per image and calls cudaStreamSynchronize after each
//     (preprocess.cu:119-127);
//   * each thread produces four horizontally adjacent output pixels and stores them as one float4 per colour plane (the 4.9 MB
//     written per 640x640 image are the traffic that matters; source bytes are gathered through L2);
//   * host-fed images are staged through a pinned ring on a dedicated copy stream (trtx_preprocess), so the H2D copy of image
//     i+1 overlaps the warp of image i instead of the reference's synchronous memcpy + copy + kernel chain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../common.h"

namespace {

constexpr int kMaxBatch = 64;

struct Image {
    const uint8_t* src;
    int w, h;
    float d2s[6];
};
struct Batch {
    Image img[kMaxBatch];
};

// one output pixel, exactly preprocess.cu:13-72
__device__ __forceinline__ void warp_pixel(const Image& im, int dx, int dy, float& c0, float& c1, float& c2) {
    const float m_x1 = im.d2s[0], m_y1 = im.d2s[1], m_z1 = im.d2s[2];
    const float m_x2 = im.d2s[3], m_y2 = im.d2s[4], m_z2 = im.d2s[5];
    const float src_x = m_x1 * dx + m_y1 * dy + m_z1 + 0.5f;
    const float src_y = m_x2 * dx + m_y2 * dy + m_z2 + 0.5f;
    const uint8_t cv = 128;
    if (src_x <= -1 || src_x >= im.w || src_y <= -1 || src_y >= im.h) {
        c0 = cv;
        c1 = cv;
        c2 = cv;
    } else {
        const int y_low = (int)floorf(src_y), x_low = (int)floorf(src_x);
        const int y_high = y_low + 1, x_high = x_low + 1;
        const uint8_t border[3] = {cv, cv, cv};
        const float ly = src_y - y_low, lx = src_x - x_low;
        const float hy = 1 - ly, hx = 1 - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const int line = im.w * 3;
        const uint8_t *v1 = border, *v2 = border, *v3 = border, *v4 = border;
        if (y_low >= 0) {
            if (x_low >= 0) v1 = im.src + (size_t)y_low * line + x_low * 3;
            if (x_high < im.w) v2 = im.src + (size_t)y_low * line + x_high * 3;
        }
        if (y_high < im.h) {
            if (x_low >= 0) v3 = im.src + (size_t)y_high * line + x_low * 3;
            if (x_high < im.w) v4 = im.src + (size_t)y_high * line + x_high * 3;
        }
        c0 = w1 * v1[0] + w2 * v2[0] + w3 * v3[0] + w4 * v4[0];
        c1 = w1 * v1[1] + w2 * v2[1] + w3 * v3[1] + w4 * v4[1];
        c2 = w1 * v1[2] + w2 * v2[2] + w3 * v3[2] + w4 * v4[2];
    }
    const float t = c2;  // bgr -> rgb
    c2 = c0;
    c0 = t;
    c0 = c0 / 255.0f;
    c1 = c1 / 255.0f;
    c2 = c2 / 255.0f;
}

__global__ __launch_bounds__(256) void letterbox_kernel(const Batch bt, float* __restrict__ dst, int dst_w, int dst_h) {
    const Image& im = bt.img[blockIdx.y];
    const int area = dst_w * dst_h;
    float* out = dst + (size_t)blockIdx.y * 3 * area;
    const int quads = (dst_w + 3) >> 2;  // four pixels of one row per thread
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= quads * dst_h) return;
    const int dy = q / quads, dx0 = (q - dy * quads) * 4;
    float r[4], g[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) warp_pixel(im, dx0 + k, dy, r[k], g[k], b[k]);
    float* p = out + (size_t)dy * dst_w + dx0;
    if (dst_w % 4 == 0) {
        *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(p + area) = make_float4(g[0], g[1], g[2], g[3]);
        *reinterpret_cast<float4*>(p + 2 * (size_t)area) = make_float4(b[0], b[1], b[2], b[3]);
    } else {
        for (int k = 0; k < 4 && dx0 + k < dst_w; ++k) {
            p[k] = r[k];
            p[k + area] = g[k];
            p[k + 2 * (size_t)area] = b[k];
        }
    }
}

// preprocess.cu:97-111: forward map (scale about the centres), then its inverse as cv::invertAffineTransform computes it for
// CV_32F (OpenCV imgproc/imgwarp.cpp: double-precision cofactors of the 2x2 part, results rounded to float).  OpenCV is an
// un-vendored dependency of the reference; this restates its published algorithm.
void inverse_map(int src_w, int src_h, int dst_w, int dst_h, float d2s[6]) {
    const float scale = std::min(dst_h / (float)src_h, dst_w / (float)src_w);
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

// host-fed path: pinned staging ring + device ring + copy stream (cuda_preprocess_init / cuda_preprocess analogue)
struct Stage {
    uint8_t* host = nullptr;
    uint8_t* dev = nullptr;
    hipEvent_t copied = nullptr, consumed = nullptr;
    bool used = false;
};
struct Preproc {
    size_t max_bytes = 0;
    std::vector<Stage> ring;
    hipStream_t copy_stream = nullptr;
    size_t next = 0;
};
Preproc* g_pre = nullptr;

}  // namespace

extern "C" void trtx_letterbox_matrix(int src_w, int src_h, int dst_w, int dst_h, float* d2s_out) { inverse_map(src_w, src_h, dst_w, dst_h, d2s_out); }

// device-resident sources: src[i] = device pointer to a tightly packed uint8 HWC BGR image of src_w[i] x src_h[i]
extern "C" int32_t trtx_letterbox_batch(const void* const* src, const int* src_w, const int* src_h, int batch, float* dst, int dst_w, int dst_h,
                                        hipStream_t stream) {
    if (!src || !src_w || !src_h || !dst || batch < 1 || dst_w < 1 || dst_h < 1) return TRTX_ERR_INVALID;
    for (int b0 = 0; b0 < batch; b0 += kMaxBatch) {
        const int nb = std::min(kMaxBatch, batch - b0);
        Batch bt;
        memset(&bt, 0, sizeof(bt));
        for (int i = 0; i < nb; ++i) {
            if (!src[b0 + i] || src_w[b0 + i] < 1 || src_h[b0 + i] < 1) return TRTX_ERR_INVALID;
            bt.img[i].src = static_cast<const uint8_t*>(src[b0 + i]);
            bt.img[i].w = src_w[b0 + i];
            bt.img[i].h = src_h[b0 + i];
            inverse_map(src_w[b0 + i], src_h[b0 + i], dst_w, dst_h, bt.img[i].d2s);
        }
        const int quads = ((dst_w + 3) / 4) * dst_h;
        hipLaunchKernelGGL(letterbox_kernel, dim3((quads + 255) / 256, nb), dim3(256), 0, stream, bt, dst + (size_t)b0 * 3 * dst_w * dst_h, dst_w,
                           dst_h);
    }
    return trtx::check_launch("trtx_letterbox_batch");
}

// cuda_preprocess_init(max_image_size) / cuda_preprocess_destroy() (preprocess.cu:129-139)
extern "C" int32_t trtx_preprocess_init(int max_image_size, int ring_depth) {
    if (g_pre || max_image_size < 1) return TRTX_ERR_STATE;
    auto* p = new Preproc();
    p->max_bytes = (size_t)max_image_size * 3;
    p->ring.resize(ring_depth < 2 ? 2 : ring_depth);
    if (hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking) != hipSuccess) return TRTX_ERR_HIP;
    for (auto& s : p->ring) {
        TRTX_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.host), p->max_bytes, hipHostMallocDefault));
        TRTX_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dev), p->max_bytes));
        TRTX_HIP_TRY(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        TRTX_HIP_TRY(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    }
    g_pre = p;
    return TRTX_OK;
}
extern "C" void trtx_preprocess_destroy() {
    if (!g_pre) return;
    (void)hipStreamSynchronize(g_pre->copy_stream);
    for (auto& s : g_pre->ring) {
        if (s.consumed) (void)hipEventSynchronize(s.consumed);
        (void)hipHostFree(s.host);
        (void)hipFree(s.dev);
        (void)hipEventDestroy(s.copied);
        (void)hipEventDestroy(s.consumed);
    }
    (void)hipStreamDestroy(g_pre->copy_stream);
    delete g_pre;
    g_pre = nullptr;
}
// cuda_batch_preprocess (preprocess.cu:119-127) for host images: every image goes pageable -> pinned slot -> device slot on the
// copy stream; the warp of the whole batch is enqueued on `stream` behind the copies.  No host synchronisation unless the ring
// wraps around onto a slot whose previous warp has not finished.
extern "C" int32_t trtx_batch_preprocess(const void* const* src_host, const int* src_w, const int* src_h, int batch, float* dst, int dst_w,
                                         int dst_h, hipStream_t stream) {
    if (!g_pre) return TRTX_ERR_STATE;
    if (!src_host || batch < 1 || batch > (int)g_pre->ring.size() || batch > kMaxBatch) return TRTX_ERR_INVALID;
    for (int i = 0; i < batch; ++i)   // refuse before any ring slot is touched
        if (!src_host[i] || src_w[i] < 1 || src_h[i] < 1 || (size_t)src_w[i] * src_h[i] * 3 > g_pre->max_bytes) return TRTX_ERR_INVALID;
    std::vector<const void*> dev(batch);
    const size_t ring = g_pre->ring.size(), first = g_pre->next;
    int touched = 0;
    // A failure half way leaves slots whose `consumed` event belongs to an older use: waiting on it would not wait for the copies
    // issued here.  So the failing call drains the copy stream itself and hands the slots back unused.
    auto give_up = [&](int32_t st) {
        (void)hipStreamSynchronize(g_pre->copy_stream);
        (void)hipGetLastError();
        for (int i = 0; i < touched; ++i) g_pre->ring[(first + i) % ring].used = false;
        return st;
    };
#define TRTX_PRE_TRY(expr)                                   \
    do {                                                     \
        if ((expr) != hipSuccess) return give_up(TRTX_ERR_HIP); \
    } while (0)
    for (int i = 0; i < batch; ++i) {
        const size_t bytes = (size_t)src_w[i] * src_h[i] * 3;
        Stage& s = g_pre->ring[g_pre->next];
        g_pre->next = (g_pre->next + 1) % ring;
        ++touched;
        if (s.used) TRTX_PRE_TRY(hipEventSynchronize(s.consumed));  // the warp that read this slot last has finished
        s.used = true;
        memcpy(s.host, src_host[i], bytes);
        TRTX_PRE_TRY(hipMemcpyAsync(s.dev, s.host, bytes, hipMemcpyHostToDevice, g_pre->copy_stream));
        TRTX_PRE_TRY(hipEventRecord(s.copied, g_pre->copy_stream));
        TRTX_PRE_TRY(hipStreamWaitEvent(stream, s.copied, 0));
        dev[i] = s.dev;
    }
    const int32_t st = trtx_letterbox_batch(dev.data(), src_w, src_h, batch, dst, dst_w, dst_h, stream);
    if (st != TRTX_OK) return give_up(st);
    for (int i = 0; i < batch; ++i) {
        Stage& s = g_pre->ring[(first + i) % ring];
        if (hipEventRecord(s.consumed, stream) != hipSuccess) {   // the warp is enqueued: wait for it, then the slots are free
            (void)hipStreamSynchronize(stream);
            return give_up(TRTX_ERR_HIP);
        }
    }
#undef TRTX_PRE_TRY
    return TRTX_OK;
}
