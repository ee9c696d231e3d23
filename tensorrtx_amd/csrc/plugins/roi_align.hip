// RoIAlign for gfx950 (MI355X): detectron2 ROIAlign(aligned=True) as the reference's RoiAlign plugin computes it
// (rcnn/RoiAlign.cu:29-153; parameters rcnn/rcnn.cpp:41-43,150-151): box * spatial_scale - 0.5, bin = roi / res,
// sampling_ratio 0 => ceil(roi / res) samples per bin and axis, bilinear samples with the [-1, size] validity window and
// edge clamp, plain average.
//
// Bilinear sampling is separable: which rows / columns a sample touches and with what weights depends on its y alone / x alone.
// Both kernels build, once per RoI, a small table per axis (res * grid entries: low index, high index, the two weights, validity)
// and then only gather:
//
//   * roi_align_nchw_f32_kernel — the plugin-ABI form (fp32 LINEAR tensors, what an IPluginV2 sees).  One workgroup per
//     (RoI, group of 32 channels): lanes own the res x res output bins (the 196 floats of a channel are one contiguous,
//     coalesced store), waves stride over the channels of the group.  Arithmetic order per sample and per bin is the
//     reference's, so the result is bit-identical to its kernel (this file is built with -ffp-contract=off).
//   * roi_align_nhwc_f16_kernel — the engine's native form (NHWC fp16 feature map in, NHWC fp16 [P][res][res][C] out, i.e.
//     exactly what the res5 convolutions consume): one wave per (RoI, bin), lanes own 8-channel chunks, every corner of every
//     sample is one fully coalesced 16-byte-per-lane read and the result one 16-byte store.  It writes 2 B per output element
//     once, where the plugin route writes 4 B and a layout pass then re-reads 4 B and writes 2 B.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct AxisSample {
    int lo, hi;      // low / high index along the axis (already clamped)
    float l, h;      // weights of the high / low neighbour: l = coord - lo, h = 1 - l
    int valid;       // coordinate inside [-1, size]
};

// RoiAlign.cu:29-57 for one axis: validity window, clamp at 0, clamp at size - 1
__device__ __forceinline__ AxisSample axis_sample(float v, int size) {
    AxisSample s;
    s.valid = !(v < -1.0 || v > size);
    if (v <= 0) v = 0;
    int lo = (int)v, hi;
    if (lo >= size - 1) {
        hi = lo = size - 1;
        v = (float)lo;
    } else {
        hi = lo + 1;
    }
    s.lo = lo;
    s.hi = hi;
    s.l = v - lo;
    s.h = (float)(1. - s.l);
    return s;
}

struct RoiGeom {
    float start_w, start_h, bin_w, bin_h;
    int grid_w, grid_h;
};

// RoiAlign.cu:101-125
__device__ __forceinline__ RoiGeom roi_geometry(const float* r, float spatial_scale, int res, int sampling_ratio) {
    RoiGeom g;
    const float roi_offset = 0.5f;
    g.start_w = r[0] * spatial_scale - roi_offset;
    g.start_h = r[1] * spatial_scale - roi_offset;
    const float end_w = r[2] * spatial_scale - roi_offset;
    const float end_h = r[3] * spatial_scale - roi_offset;
    const float roi_w = end_w - g.start_w, roi_h = end_h - g.start_h;
    g.bin_h = roi_h / (float)res;
    g.bin_w = roi_w / (float)res;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / res);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / res);
    return g;
}
// sample coordinate of (bin p, sample i) along an axis (RoiAlign.cu:131-136)
__device__ __forceinline__ float sample_coord(float start, int p, float bin, int i, int grid) {
    return start + p * bin + (float)(i + .5f) * bin / (float)grid;
}

constexpr int kTab = 1024;      // table entries per axis held in LDS (res * grid); larger RoIs recompute per sample
constexpr int kChanGroup = 32;  // channels per workgroup (plugin-ABI kernel)

__global__ __launch_bounds__(256) void roi_align_nchw_f32_kernel(const float* __restrict__ features, const float* __restrict__ rois,
                                                                 float spatial_scale, int channels, int height, int width, int res,
                                                                 int sampling_ratio, int num_proposals, float* __restrict__ top) {
    __shared__ AxisSample s_y[kTab], s_x[kTab];
    const int n = blockIdx.x;                 // proposal index across the batch
    const int c0 = blockIdx.y * kChanGroup;
    const int b = n / num_proposals;
    const RoiGeom g = roi_geometry(rois + (size_t)n * 4, spatial_scale, res, sampling_ratio);
    const bool tab = g.grid_h > 0 && g.grid_w > 0 && res * g.grid_h <= kTab && res * g.grid_w <= kTab;
    if (tab) {
        for (int e = threadIdx.x; e < res * g.grid_h; e += blockDim.x)
            s_y[e] = axis_sample(sample_coord(g.start_h, e / g.grid_h, g.bin_h, e % g.grid_h, g.grid_h), height);
        for (int e = threadIdx.x; e < res * g.grid_w; e += blockDim.x)
            s_x[e] = axis_sample(sample_coord(g.start_w, e / g.grid_w, g.bin_w, e % g.grid_w, g.grid_w), width);
    }
    __syncthreads();
    const int bins = res * res;
    const float count = g.grid_h * g.grid_w;  // float(int product), as the reference
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cend = min(c0 + kChanGroup, channels);
    // wave w takes channels c0 + w, c0 + w + 4, ...; its lanes sweep the bins of the channel in passes of 64
    for (int c = c0 + wave; c < cend; c += 4) {
        const float* plane = features + ((size_t)b * channels + c) * height * width;
        float* dst = top + ((size_t)n * channels + c) * bins;
        for (int bin = lane; bin < bins; bin += 64) {
            const int ph = bin / res, pw = bin - ph * res;
            float acc = 0.f;
            for (int iy = 0; iy < g.grid_h; ++iy) {
                const AxisSample sy = tab ? s_y[ph * g.grid_h + iy] : axis_sample(sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h), height);
                const float* row_lo = plane + (size_t)sy.lo * width;
                const float* row_hi = plane + (size_t)sy.hi * width;
                for (int ix = 0; ix < g.grid_w; ++ix) {
                    const AxisSample sx = tab ? s_x[pw * g.grid_w + ix] : axis_sample(sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w), width);
                    float val = 0.f;
                    if (sy.valid && sx.valid) {
                        const float v1 = row_lo[sx.lo], v2 = row_lo[sx.hi], v3 = row_hi[sx.lo], v4 = row_hi[sx.hi];
                        const float w1 = sy.h * sx.h, w2 = sy.h * sx.l, w3 = sy.l * sx.h, w4 = sy.l * sx.l;
                        val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                    }
                    acc += val;
                }
            }
            acc /= count;
            dst[bin] = acc;
        }
    }
}

// one wave per (RoI, bin); lanes own 8-channel chunks.  features NHWC fp16 [batch][height][width][ld_in], out NHWC fp16
// [batch * num_proposals][ores][ores][ld_out] with ores = (res - 1) / step + 1: output bin (oh, ow) is bin (oh * step, ow * step) of the
// res x res grid.  step = 1 is the operator itself; step = 2 is what a 1x1 STRIDE-2 consumer reads of it (res5.0's conv1 and shortcut,
// rcnn/backbone.hpp:9,110-117 STRIDE_IN_1X1): the other three quarters of the bins are never computed or written.
__global__ __launch_bounds__(256) void roi_align_nhwc_f16_kernel(const _Float16* __restrict__ features, int ld_in, const float* __restrict__ rois,
                                                                 float spatial_scale, int channels, int height, int width, int res,
                                                                 int sampling_ratio, int num_proposals, _Float16* __restrict__ out, int ld_out, int step) {
    const int n = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int obin = blockIdx.x * 4 + wave;
    const int ores = (res - 1) / step + 1;
    if (obin >= ores * ores) return;  // whole waves
    const int oh = obin / ores, ow = obin - oh * ores;
    const int ph = oh * step, pw = ow * step;
    const int b = n / num_proposals;
    const RoiGeom g = roi_geometry(rois + (size_t)n * 4, spatial_scale, res, sampling_ratio);
    const _Float16* img = features + (size_t)b * height * width * ld_in;
    _Float16* dst = out + (((size_t)n * ores + oh) * ores + ow) * ld_out;
    const float inv_count = 1.0f / (float)(g.grid_h * g.grid_w);
    const int chunks = channels >> 3;
    for (int ck = lane; ck < chunks; ck += 64) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const AxisSample sy = axis_sample(sample_coord(g.start_h, ph, g.bin_h, iy, g.grid_h), height);  // wave-uniform: scalar work
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const AxisSample sx = axis_sample(sample_coord(g.start_w, pw, g.bin_w, ix, g.grid_w), width);
                if (!(sy.valid && sx.valid)) continue;
                const float w1 = sy.h * sx.h, w2 = sy.h * sx.l, w3 = sy.l * sx.h, w4 = sy.l * sx.l;
                const half8 v1 = *reinterpret_cast<const half8*>(img + ((size_t)sy.lo * width + sx.lo) * ld_in + ck * 8);
                const half8 v2 = *reinterpret_cast<const half8*>(img + ((size_t)sy.lo * width + sx.hi) * ld_in + ck * 8);
                const half8 v3 = *reinterpret_cast<const half8*>(img + ((size_t)sy.hi * width + sx.lo) * ld_in + ck * 8);
                const half8 v4 = *reinterpret_cast<const half8*>(img + ((size_t)sy.hi * width + sx.hi) * ld_in + ck * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += w1 * (float)v1[e] + w2 * (float)v2[e] + w3 * (float)v3[e] + w4 * (float)v4[e];
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(acc[e] * inv_count);
        *reinterpret_cast<half8*>(dst + ck * 8) = o;
    }
}

}  // namespace

// roiAlign (rcnn/RoiAlign.cu:83-182), plugin ABI: boxes [batch][P][4], features [batch][C][fh][fw] -> out [batch][P][C][res][res], fp32
extern "C" int32_t trtx_roi_align(int batch, const float* boxes, const float* features, int pooler_resolution, float spatial_scale,
                                  int sampling_ratio, int num_proposals, int channels, int feature_h, int feature_w, float* out,
                                  hipStream_t stream) {
    if (!boxes || !features || !out || batch < 1 || pooler_resolution < 1 || num_proposals < 1 || channels < 1 || feature_h < 1 || feature_w < 1)
        return TRTX_ERR_INVALID;
    const dim3 grid((unsigned)(batch * num_proposals), (unsigned)((channels + kChanGroup - 1) / kChanGroup));
    hipLaunchKernelGGL(roi_align_nchw_f32_kernel, grid, dim3(256), 0, stream, features, boxes, spatial_scale, channels, feature_h, feature_w,
                       pooler_resolution, sampling_ratio, num_proposals, out);
    return trtx::check_launch("trtx_roi_align");
}

// engine-native form: features NHWC fp16 [batch][fh][fw][ld_in] (channels % 8 == 0, 16-byte aligned), out NHWC fp16
// [batch * P][res][res][ld_out]
extern "C" int32_t trtx_roi_align_nhwc_f16(int batch, const float* boxes, const void* features, int ld_in, int pooler_resolution,
                                           float spatial_scale, int sampling_ratio, int num_proposals, int channels, int feature_h,
                                           int feature_w, void* out, int ld_out, hipStream_t stream) {
    return trtx_roi_align_nhwc_f16_strided(batch, boxes, features, ld_in, pooler_resolution, spatial_scale, sampling_ratio, num_proposals, channels,
                                           feature_h, feature_w, out, ld_out, 1, stream);
}

// ... every bin_step-th bin per axis only: out NHWC fp16 [batch * P][ores][ores][ld_out], ores = (res - 1) / bin_step + 1
extern "C" int32_t trtx_roi_align_nhwc_f16_strided(int batch, const float* boxes, const void* features, int ld_in, int pooler_resolution,
                                                   float spatial_scale, int sampling_ratio, int num_proposals, int channels, int feature_h,
                                                   int feature_w, void* out, int ld_out, int bin_step, hipStream_t stream) {
    if (!boxes || !features || !out || batch < 1 || pooler_resolution < 1 || num_proposals < 1 || channels < 8 || feature_h < 1 || feature_w < 1 ||
        bin_step < 1)
        return TRTX_ERR_INVALID;
    if (channels % 8 || ld_in % 8 || ld_out % 8 || (reinterpret_cast<uintptr_t>(features) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return TRTX_ERR_UNSUPPORTED;
    const int ores = (pooler_resolution - 1) / bin_step + 1;
    const int bins = ores * ores;
    const dim3 grid((unsigned)((bins + 3) / 4), (unsigned)(batch * num_proposals));
    hipLaunchKernelGGL(roi_align_nhwc_f16_kernel, grid, dim3(256), 0, stream, static_cast<const _Float16*>(features), ld_in, boxes, spatial_scale,
                       channels, feature_h, feature_w, pooler_resolution, sampling_ratio, num_proposals, static_cast<_Float16*>(out), ld_out, bin_step);
    return trtx::check_launch("trtx_roi_align_nhwc_f16");
}
