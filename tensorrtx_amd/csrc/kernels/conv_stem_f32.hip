// First layer of an fp32 engine: the fp32 NCHW network input (1..4 channels, the image) convolved straight into the NHWC fp32 tensor the MFMA layers read -
// no layout pass (for YOLOv8n b32 the to_nhwc copy alone was 126 us and the 3 -> 16 stem 203 us on the implicit-GEMM kernel, whose 16-channel k-step is
// 3x padding on a 27-long reduction; profiles/r05_f32_first_run.txt).  yolov8/src/model.cpp:115 (conv 3 -> 16, 3x3 / 2), resnet/resnet50.cpp:176 (7x7 / 2).
//
// Bound: HBM.  Per image at 640 x 640: 4.9 MB read once + 6.6 MB written; 88 MFLOP of fp32 FMA - 0.9 us of the chip's 157 TFLOP/s vector rate against
// 1.4 us of HBM time.  So: plain VALU, no LDS, no MFMA.  One lane = one output pixel x 16 output channels (blockIdx.y walks the channel groups); the lanes of
// a wave are consecutive pixels of an output row, so a tap's input load is one row segment (stride-2 gather served by the L1 line), the 16 weights of a tap
// are wave-uniform (scalar loads, used as the SGPR operand of v_fmac_f32), and a lane's result is 64 contiguous bytes - the wave writes 4 KB in a row.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {

constexpr int kCo = 16;   // output channels per lane

__device__ __forceinline__ float stem_act(float v, int act, float alpha) {
    if (act == ACT_NONE) return v;
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU || act == ACT_SIGMOID) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));   // as conv_epilogue_f32
        return act == ACT_SILU ? v * sg : sg;
    }
    if (act == ACT_LEAKY) return v > 0.f ? v : v * alpha;
    if (act == ACT_TANH) return tanhf(v);
    return mish_ref(v);
}

// KS: filter size when square and known (3, 7), 0 = runtime kh x kw
template <int KS>
__global__ __launch_bounds__(256) void conv_stem_f32_kernel(const ConvArgs p) {
    const float* __restrict__ in = static_cast<const float*>(p.in);
    const float* __restrict__ w = static_cast<const float*>(p.wgt);   // [(c*kh + r)*kw + q][Cout]
    float* __restrict__ out = static_cast<float*>(p.out);
    const int kh = KS ? KS : p.kh, kw = KS ? KS : p.kw;
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * kCo;
    if (m >= p.M) return;
    const int wo = (int)(m % p.Wo);
    const long t = m / p.Wo;
    const int ho = (int)(t % p.Ho);
    const long n = t / p.Ho;
    const int hi0 = ho * p.stride_h - p.pad_h, wi0 = wo * p.stride_w - p.pad_w;
    float acc[kCo];
#pragma unroll
    for (int j = 0; j < kCo; ++j) acc[j] = p.bias ? p.bias[co0 + j] : 0.f;
    for (int c = 0; c < p.Cin; ++c) {
        const float* plane = in + (n * p.Cin + c) * (long)p.H * p.W;
        auto tap = [&](int r, int q) {
            const int hi = hi0 + r, wi = wi0 + q;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const float x = plane[ok ? (long)hi * p.W + wi : 0];   // unconditional load from a clamped address, masked afterwards
            const float xv = ok ? x : 0.f;
            const float* wt = w + (size_t)((c * kh + r) * kw + q) * p.Cout + co0;   // wave-uniform
#pragma unroll
            for (int j = 0; j < kCo; ++j) acc[j] = fmaf(xv, wt[j], acc[j]);
        };
        if constexpr (KS != 0) {
#pragma unroll
            for (int r = 0; r < KS; ++r)
#pragma unroll
                for (int q = 0; q < KS; ++q) tap(r, q);
        } else {
            for (int r = 0; r < kh; ++r)
                for (int q = 0; q < kw; ++q) tap(r, q);
        }
    }
    float* o = out + (size_t)m * p.ld_out + co0;
#pragma unroll
    for (int j = 0; j < kCo; j += 4) {
        float4 v;
        v.x = stem_act(acc[j], p.act1, p.alpha1);
        v.y = stem_act(acc[j + 1], p.act1, p.alpha1);
        v.z = stem_act(acc[j + 2], p.act1, p.alpha1);
        v.w = stem_act(acc[j + 3], p.act1, p.alpha1);
        *reinterpret_cast<float4*>(o + j) = v;
    }
}

}  // namespace

bool conv_stem_f32_supported(const ConvArgs& a) {
    return a.Cin >= 1 && a.Cin <= 4 && a.Cout % kCo == 0 && a.Cout <= 256 && a.groups == 1 && a.dil_h == 1 && a.dil_w == 1 && !a.residual && a.act2 == ACT_NONE &&
           (double)a.N * a.Cin * a.H * a.W < 2.0e9;
}

// in: fp32 NCHW [N][Cin][H][W]; wgt: fp32 [kh*kw*Cin (c, r, q order)][Cout]; out: NHWC fp32 (ld_out % 4 == 0, 16-byte aligned)
int32_t conv_stem_nchw_f32_out_f32(const ConvArgs& a, hipStream_t s) {
    if (!conv_stem_f32_supported(a) || a.ld_out % 4 || (reinterpret_cast<uintptr_t>(a.out) & 15)) return TRTX_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(((long)a.M + 255) / 256), a.Cout / kCo), block(256);
    if (a.kh == 3 && a.kw == 3) hipLaunchKernelGGL(conv_stem_f32_kernel<3>, grid, block, 0, s, a);
    else if (a.kh == 7 && a.kw == 7) hipLaunchKernelGGL(conv_stem_f32_kernel<7>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(conv_stem_f32_kernel<0>, grid, block, 0, s, a);
    return check_launch("conv_stem_nchw_f32_out_f32");
}

}  // namespace trtx
