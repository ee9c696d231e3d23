// Internal helpers shared by the HIP kernels and the runtime (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "trtx_hip.h"

namespace trtx {

#if defined(__HIPCC__)
// fp32 -> fp16 with the fp32 value materialised first.  Without the (empty) asm the compiler fuses "a * b" with the conversion into
// v_fma_mixlo_f16 for SOME elements of an unrolled epilogue (one rounding) and keeps v_mul_f32 + v_cvt_f16_f32 for others (two
// roundings): the same pixel then differs by one fp16 ulp depending on which fragment of a tile - i.e. which batch position - it
// lands in (tests/test_gpu_engine.py::test_yolov8n_fp16_engine_640_batch32_the_bench_configuration).
__device__ __forceinline__ _Float16 round_to_half(float v) {
    asm("" : "+v"(v));
    return (_Float16)v;
}
// Mish as the reference's plugin kernel computes it (yolov4/mish.cu:113-135): softplus with its threshold of 20 on both sides,
// tanh spelled 2 / (1 + exp(-2y)) - 1, the accurate expf / logf.  No multiply-add pairs: immune to -ffp-contract.
__device__ __forceinline__ float mish_ref(float x) {
    const float sp = x > 20.f ? x : (x < -20.f ? expf(x) : logf(expf(x) + 1.f));
    return x * (2.f / (1.f + expf(-2.f * sp)) - 1.f);
}
#endif

constexpr int kYoloDetFloats = 90;  // sizeof(Detection)/4, yolov8/include/types.h:4-12

inline size_t align_up(size_t v, size_t a) {
    return (v + a - 1) / a * a;
}

// Launch errors are reported, never swallowed: the product path must fail loudly.
inline int32_t check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "[trtx_hip] %s: %s\n", what, hipGetErrorString(e));
        return TRTX_ERR_HIP;
    }
    return TRTX_OK;
}

#define TRTX_HIP_TRY(expr)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            fprintf(stderr, "[trtx_hip] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr,            \
                    hipGetErrorString(e__));                                                     \
            return TRTX_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

// Monotone map float -> uint32 (ascending).  -0.0f is folded onto +0.0f so that keys compare the
// way IEEE '<' / '==' do.
__host__ __device__ inline uint32_t ord_f32(float f) {
    f = f + 0.0f;
    union {
        float f;
        uint32_t u;
    } c;
    c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}

}  // namespace trtx
