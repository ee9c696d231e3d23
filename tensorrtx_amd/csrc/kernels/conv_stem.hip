// First-layer ("stem") convolution for gfx950: reads the network input as the caller hands it over —
// fp32 NCHW (LINEAR) with 1..4 channels — and writes NHWC fp16 with bias(+folded BN) and activation fused,
// so the layout/dtype conversion pass over the largest tensor of the network disappears.
// (YOLOv8 model.0: 3->16 3x3/2, yolov8/src/model.cpp:115; ResNet-50 conv1: 3->64 7x7/2, resnet50.cpp:165-170.)
//
// With K = kh*kw*Cin = 27..147 and Cout <= 64 this layer is far below the MFMA ridge: it is bound by the HBM
// read of the fp32 image and the fp16 store.  One thread per output pixel keeps COUT fp32 accumulators in
// registers; lanes walk consecutive output columns so the NCHW loads are coalesced along W and the NHWC
// store is one contiguous COUT*2-byte run per lane.  Weights ([tap][COUT] fp32) are broadcast from LDS as
// float4.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float stem_act(float v, int act, float alpha) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case ACT_SILU: return v / (1.0f + __expf(-v));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void conv_stem_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // [K][COUT]
    const float* __restrict__ w = static_cast<const float*>(p.wgt);
    const int K = p.kh * p.kw * p.Cin;
    for (int i = threadIdx.x; i < K * COUT; i += 256) s_w[i] = w[i];
    __syncthreads();
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= p.M) return;
    const int wo = (int)(m % p.Wo);
    const long t = m / p.Wo;
    const int ho = (int)(t % p.Ho);
    const long n = t / p.Ho;
    const float* __restrict__ in = static_cast<const float*>(p.in) + n * p.Cin * p.H * p.W;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = p.bias ? p.bias[c] : 0.f;
    const int hi0 = ho * p.stride_h - p.pad_h, wi0 = wo * p.stride_w - p.pad_w;
    int tap = 0;
    for (int c = 0; c < p.Cin; ++c) {
        const float* plane = in + (long)c * p.H * p.W;
        for (int r = 0; r < p.kh; ++r) {
            const int hi = hi0 + r;
            const bool hok = (unsigned)hi < (unsigned)p.H;
            for (int q = 0; q < p.kw; ++q, ++tap) {
                const int wi = wi0 + q;
                float x = 0.f;
                if (hok && (unsigned)wi < (unsigned)p.W) x = plane[(long)hi * p.W + wi];
                const float4* wr = reinterpret_cast<const float4*>(s_w + tap * COUT);
#pragma unroll
                for (int c4 = 0; c4 < COUT / 4; ++c4) {
                    const float4 wv = wr[c4];
                    acc[c4 * 4 + 0] = fmaf(x, wv.x, acc[c4 * 4 + 0]);
                    acc[c4 * 4 + 1] = fmaf(x, wv.y, acc[c4 * 4 + 1]);
                    acc[c4 * 4 + 2] = fmaf(x, wv.z, acc[c4 * 4 + 2]);
                    acc[c4 * 4 + 3] = fmaf(x, wv.w, acc[c4 * 4 + 3]);
                }
            }
        }
    }
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out) + m * p.ld_out;
#pragma unroll
    for (int c8 = 0; c8 < COUT / 8; ++c8) {
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)stem_act(acc[c8 * 8 + e], p.act1, p.alpha1);
        *reinterpret_cast<half8*>(out + c8 * 8) = v;
    }
}

template <int COUT>
void launch(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)a.kh * a.kw * a.Cin * COUT * sizeof(float);
    hipLaunchKernelGGL(conv_stem_kernel<COUT>, dim3((unsigned)((a.M + 255) / 256)), dim3(256), lds, s, a);
}

}  // namespace

bool conv_stem_supported(const ConvArgs& a) {
    const bool cout_ok = a.Cout == 8 || a.Cout == 16 || a.Cout == 32 || a.Cout == 64;
    return cout_ok && a.Cin >= 1 && a.Cin <= 4 && a.groups == 1 && a.dil_h == 1 && a.dil_w == 1 && !a.residual &&
           a.act2 == ACT_NONE && (size_t)a.kh * a.kw * a.Cin * a.Cout * 4 <= 48 * 1024;
}

// in: fp32 NCHW [N][Cin][H][W]; wgt: fp32 [kh*kw*Cin (c,r,q order)][Cout]; out: NHWC fp16 (ld_out % 8 == 0, 16-B aligned)
int32_t conv_stem_nchw_f32(const ConvArgs& a, hipStream_t s) {
    if (!conv_stem_supported(a) || a.ld_out % 8 || (reinterpret_cast<uintptr_t>(a.out) & 15)) return TRTX_ERR_UNSUPPORTED;
    switch (a.Cout) {
        case 8: launch<8>(a, s); break;
        case 16: launch<16>(a, s); break;
        case 32: launch<32>(a, s); break;
        case 64: launch<64>(a, s); break;
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return check_launch("conv_stem_nchw_f32");
}

}  // namespace trtx
