// INT8 support kernels for gfx950: calibration statistics over NHWC fp16 tensors (per-tensor |x| maximum and 2048-bin
// histogram, what IInt8EntropyCalibrator2 needs - yolov8/src/calibrator.cpp:9-74 feeds the batches, TensorRT collects the
// statistics; this is that collector) and the int8 nearest-neighbour resize with requantisation.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int kBins = 8192;   // == runtime/int8.h kCalibBins (32 KB of LDS per workgroup)

// max |x| over [pixels][C] (channel stride ld), C % 8 == 0: one atomicMax of the float bits (non-negative floats order as ints)
__global__ __launch_bounds__(256) void absmax_f16_kernel(const _Float16* __restrict__ x, long pixels, int C, int ld, unsigned* __restrict__ out) {
    const int chunks = C >> 3;
    const long total = pixels * chunks;
    float m = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long px = i / chunks;
        const int ck = (int)(i - px * chunks);
        const half8 v = *reinterpret_cast<const half8*>(x + px * ld + ck * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = fabsf((float)v[e]);
            m = (a > m && a < 6.0e4f) ? a : m;  // inf / NaN do not define a range
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]))));
}

// hist[bin(|x|)] += 1 with bin = |x| / range * 2048 (values beyond the range go to the last bin); LDS-private histogram per block
__global__ __launch_bounds__(256) void hist_f16_kernel(const _Float16* __restrict__ x, long pixels, int C, int ld, float inv_bin,
                                                      unsigned long long* __restrict__ hist) {
    __shared__ unsigned s_h[kBins];
    for (int i = threadIdx.x; i < kBins; i += 256) s_h[i] = 0;
    __syncthreads();
    const int chunks = C >> 3;
    const long total = pixels * chunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long px = i / chunks;
        const int ck = (int)(i - px * chunks);
        const half8 v = *reinterpret_cast<const half8*>(x + px * ld + ck * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = fabsf((float)v[e]);
            if (!(a < 6.0e4f)) continue;
            int b = (int)(a * inv_bin);
            b = b < kBins ? b : kBins - 1;
            atomicAdd(&s_h[b], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += 256)
        if (s_h[i]) atomicAdd(&hist[i], (unsigned long long)s_h[i]);
}

// nearest-neighbour resize of an int8 NHWC tensor (C % 16 == 0) with requantisation q_out = round(q_in * ratio), ratio = s_in / s_out
__global__ __launch_bounds__(256) void resize_nearest_i8_kernel(const int8_t* __restrict__ in, int8_t* __restrict__ out, int N, int H, int W, int C,
                                                               int ld_in, int Ho, int Wo, int ld_out, float ratio) {
    const int chunks = C >> 4;
    const long total = (long)N * Ho * Wo * chunks;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ck = (int)(i % chunks);
    long t = i / chunks;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int hi = min((int)((long)ho * H / Ho), H - 1), wi = min((int)((long)wo * W / Wo), W - 1);
    const int4 v = *reinterpret_cast<const int4*>(in + (((size_t)n * H + hi) * W + wi) * ld_in + ck * 16);
    int4 o = v;
    if (ratio != 1.0f) {
        const int8_t* src = reinterpret_cast<const int8_t*>(&v);
        int8_t* dst = reinterpret_cast<int8_t*>(&o);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float q = rintf((float)src[e] * ratio);
            q = q > 127.f ? 127.f : (q < -127.f ? -127.f : q);
            dst[e] = (int8_t)(int)q;
        }
    }
    *reinterpret_cast<int4*>(out + (((size_t)n * Ho + ho) * Wo + wo) * ld_out + ck * 16) = o;
}

}  // namespace

int32_t nhwc_absmax_f16(const void* x, long pixels, int C, int ld, unsigned* out_bits, hipStream_t s) {
    if (C % 8 || ld % 8) return TRTX_ERR_UNSUPPORTED;
    const long total = pixels * (C / 8);
    const int blocks = (int)std::min<long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(absmax_f16_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, static_cast<const _Float16*>(x), pixels, C, ld, out_bits);
    return check_launch("nhwc_absmax_f16");
}

int32_t nhwc_hist_f16(const void* x, long pixels, int C, int ld, float range, unsigned long long* hist, hipStream_t s) {
    if (C % 8 || ld % 8 || !(range > 0.f)) return TRTX_ERR_UNSUPPORTED;
    const long total = pixels * (C / 8);
    const int blocks = (int)std::min<long>((total + 255) / 256, 1024);
    hipLaunchKernelGGL(hist_f16_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, static_cast<const _Float16*>(x), pixels, C, ld,
                       (float)kBins / range, hist);
    return check_launch("nhwc_hist_f16");
}

int32_t nhwc_resize_nearest_i8(const void* in, void* out, int N, int H, int W, int C, int ld_in, int Ho, int Wo, int ld_out, float ratio,
                               hipStream_t s) {
    if (C % 16 || ld_in % 16 || ld_out % 16) return TRTX_ERR_UNSUPPORTED;
    const long total = (long)N * Ho * Wo * (C / 16);
    hipLaunchKernelGGL(resize_nearest_i8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const int8_t*>(in),
                       static_cast<int8_t*>(out), N, H, W, C, ld_in, Ho, Wo, ld_out, ratio);
    return check_launch("nhwc_resize_nearest_i8");
}

}  // namespace trtx
