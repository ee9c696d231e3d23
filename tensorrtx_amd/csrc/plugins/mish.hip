// "Mish_TRT" as a stand-alone launch (reference yolov4/mish.{h,cu}; the built-in plugin of plugins/builtin_plugins.cpp calls it when no
// convolution absorbed the activation): out[i] = x * tanh(softplus(x)), softplus with the reference's threshold of 20 on both sides
// (mish.cu:113-117), tanh spelled 2 / (1 + exp(-2y)) - 1 (mish.cu:111).
//
// This file is built with -ffp-contract=off like every plugin kernel (csrc/Makefile PLUGIN_FLAGS): log(exp(x) + 1) for x < 0 turns ONE ulp
// of expf into up to 64 ulp of the result, and whether the device math library's expf / logf fuse their internal multiply-adds follows the
// translation unit's contraction mode.  The reference kernel compiled the same way (oracle/ref_build.py) then agrees BIT FOR BIT
// (tests/test_ref_pinning.py::test_builtin_mish_equals_the_reference_kernel_bit_for_bit; under the kernels' default contraction mode
// 2 % of the elements differed, by up to 64 ulp).  One 16-byte access per lane where the count allows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"

namespace {

__global__ __launch_bounds__(256) void mish_f32_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n4, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        float4 o;
        o.x = trtx::mish_ref(v.x);
        o.y = trtx::mish_ref(v.y);
        o.z = trtx::mish_ref(v.z);
        o.w = trtx::mish_ref(v.w);
        reinterpret_cast<float4*>(out)[i] = o;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = trtx::mish_ref(in[i]);
}

}  // namespace

// mish_kernel of the reference's Mish_TRT plugin (yolov4/mish.cu:119-141) over n fp32 values
extern "C" int32_t trtx_mish(const float* in, float* out, size_t n, trtx_stream_t stream) {
    if ((!in || !out) && n) return TRTX_ERR_INVALID;
    if (!n) return TRTX_OK;
    const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    const size_t work = n4 + (n - 4 * n4);
    const unsigned blocks = (unsigned)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
    hipLaunchKernelGGL(mish_f32_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, n4, n);
    return trtx::check_launch("trtx_mish");
}
