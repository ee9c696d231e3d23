// Test support (not on any product path): fills every CU's LDS with fp16 NaN patterns.  LDS is not cleared between kernels, so a kernel that reads LDS
// bytes it never wrote sees whatever the previous kernel left; the multi-context stress tests poison it between rounds (tests/test_gpu_multi_context.py).
#include <hip/hip_runtime.h>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_words[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_words[i] = 0x7fff7fffu;
    __syncthreads();
    if (lds_words[(threadIdx.x * 37) % (160 * 1024 / 4)] == 1u) *sink = 1u;  // keeps the stores alive
}
}  // namespace

int32_t poison_lds(unsigned* device_word, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_done = true;
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 160 * 1024, s, device_word);  // one workgroup per CU at a time, four rounds
    return check_launch("poison_lds");
}

}  // namespace trtx
