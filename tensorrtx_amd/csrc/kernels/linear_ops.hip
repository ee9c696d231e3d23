// fp32 kernels for tensors kept in LINEAR (row-major, batch outermost) layout — the small "tail" ops
// of the reference graphs: addShuffle / addSlice / addConcatenation on reshaped heads, addSoftMax,
// addMatrixMultiply + addConstant (lenet/lenet.cpp:86-131), DFL (yolov8/src/block.cpp:239-257),
// addScale on non-image tensors, addReduce.
//
// All HBM/latency-bound and tiny next to the convolutions; written as straightforward grid-stride
// gathers with the contiguous output index on the lanes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256 * 8;

inline int grid_for(long work) {
    long b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    return (int)(b > kMaxBlocks ? kMaxBlocks : b);
}

__device__ __forceinline__ float act_f(float v, int act, float alpha) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        case ACT_MISH: return mish_ref(v);
        default: return v;
    }
}

__device__ __forceinline__ float ew_f(float a, float b, int op) {
    switch (op) {
        case EW_SUM: return a + b;
        case EW_PROD: return a * b;
        case EW_MAX: return a > b ? a : b;
        case EW_MIN: return a < b ? a : b;
        case EW_SUB: return a - b;
        case EW_DIV: return a / b;
        case EW_POW: return powf(a, b);
        default: return a;
    }
}

__device__ __forceinline__ void decompose(long i, const StridedView& v, long& off1, long& off2) {
    off1 = 0;
    off2 = 0;
#pragma unroll
    for (int d = 5; d >= 0; --d) {
        if (d < v.rank) {
            const long idx = i % v.shape[d];
            i /= v.shape[d];
            off1 += idx * v.stride_in[d];
            off2 += idx * v.stride_in2[d];
        }
    }
}

__global__ void gather_kernel(const float* __restrict__ in, float* __restrict__ out, const StridedView v, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long o1, o2;
        decompose(i, v, o1, o2);
        out[i] = in[o1];
    }
}

__global__ void scatter_kernel(const float* __restrict__ in, float* __restrict__ out, const StridedView v,
                               long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long o1, o2;
        decompose(i, v, o1, o2);
        out[o1] = in[i];
    }
}

__global__ void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int op,
                          const StridedView v, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long o1, o2;
        decompose(i, v, o1, o2);
        out[i] = ew_f(a[o1], b[o2], op);
    }
}

__global__ void act_kernel(const float* __restrict__ in, float* __restrict__ out, int act, float alpha, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = act_f(in[i], act, alpha);
}

// one thread per (outer, inner) pair, sequential over the (short) softmax axis
__global__ void softmax_kernel(const float* __restrict__ in, float* __restrict__ out, long outer, long axis,
                               long inner) {
    const long total = outer * inner;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long o = i / inner, in_i = i - o * inner;
        const float* p = in + o * axis * inner + in_i;
        float* q = out + o * axis * inner + in_i;
        float mx = -INFINITY;
        for (long a = 0; a < axis; ++a) mx = fmaxf(mx, p[a * inner]);
        float sum = 0.f;
        for (long a = 0; a < axis; ++a) sum += expf(p[a * inner] - mx);
        const float inv = 1.0f / sum;
        for (long a = 0; a < axis; ++a) q[a * inner] = expf(p[a * inner] - mx) * inv;
    }
}

__global__ void matmul_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                              int batch, int M, int N, int K, int ta, int tb, long bsA, long bsB) {
    const long total = (long)batch * M * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const long t = i / N;
        const int m = (int)(t % M);
        const long b = t / M;
        const float* a = A + b * bsA;
        const float* bb = B + b * bsB;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const float x = ta ? a[(long)k * M + m] : a[(long)m * K + k];
            const float y = tb ? bb[(long)n * K + k] : bb[(long)k * N + n];
            acc = fmaf(x, y, acc);
        }
        C[i] = acc;
    }
}

__global__ void reduce_kernel(const float* __restrict__ in, float* __restrict__ out, int op, long outer, long axis,
                              long inner) {
    const long total = outer * inner;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long o = i / inner, in_i = i - o * inner;
        const float* p = in + o * axis * inner + in_i;
        float acc = op == 2 ? -INFINITY : 0.f;
        for (long a = 0; a < axis; ++a) {
            const float v = p[a * inner];
            acc = op == 2 ? fmaxf(acc, v) : acc + v;
        }
        if (op == 1) acc /= (float)axis;
        out[i] = acc;
    }
}

// mode 0 uniform, 1 per-channel: y = (x*scale + shift)^power
__global__ void scale_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ scale,
                             const float* __restrict__ shift, const float* __restrict__ power, int mode, long outer,
                             long C, long inner) {
    const long total = outer * C * inner;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long c = mode == 1 ? (i / inner) % C : 0;
        float v = in[i];
        v = v * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
        if (power && power[c] != 1.0f) v = powf(v, power[c]);
        out[i] = v;
    }
}

inline long total_of(const StridedView& v) {
    long t = 1;
    for (int d = 0; d < v.rank; ++d) t *= v.shape[d];
    return t;
}

}  // namespace

int32_t lin_gather(const float* in, float* out, const StridedView& v, hipStream_t s) {
    const long total = total_of(v);
    if (total == 0) return TRTX_OK;
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(total)), dim3(kThreads), 0, s, in, out, v, total);
    return check_launch("lin_gather");
}

int32_t lin_scatter(const float* in, float* out, const StridedView& v, hipStream_t s) {
    const long total = total_of(v);
    if (total == 0) return TRTX_OK;
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(total)), dim3(kThreads), 0, s, in, out, v, total);
    return check_launch("lin_scatter");
}

int32_t lin_elementwise(const float* a, const float* b, float* out, int op, const StridedView& v, hipStream_t s) {
    const long total = total_of(v);
    if (total == 0) return TRTX_OK;
    hipLaunchKernelGGL(ew_kernel, dim3(grid_for(total)), dim3(kThreads), 0, s, a, b, out, op, v, total);
    return check_launch("lin_elementwise");
}

int32_t lin_activation(const float* in, float* out, int act, float alpha, long n, hipStream_t s) {
    if (n == 0) return TRTX_OK;
    hipLaunchKernelGGL(act_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, in, out, act, alpha, n);
    return check_launch("lin_activation");
}

int32_t lin_softmax(const float* in, float* out, long outer, long axis, long inner, hipStream_t s) {
    hipLaunchKernelGGL(softmax_kernel, dim3(grid_for(outer * inner)), dim3(kThreads), 0, s, in, out, outer, axis,
                       inner);
    return check_launch("lin_softmax");
}

int32_t lin_matmul(const float* A, const float* B, float* C, int batch, int M, int N, int K, int ta, int tb, long bsA,
                   long bsB, hipStream_t s) {
    hipLaunchKernelGGL(matmul_kernel, dim3(grid_for((long)batch * M * N)), dim3(kThreads), 0, s, A, B, C, batch, M, N,
                       K, ta, tb, bsA, bsB);
    return check_launch("lin_matmul");
}

int32_t lin_reduce(const float* in, float* out, int op, long outer, long axis, long inner, hipStream_t s) {
    hipLaunchKernelGGL(reduce_kernel, dim3(grid_for(outer * inner)), dim3(kThreads), 0, s, in, out, op, outer, axis,
                       inner);
    return check_launch("lin_reduce");
}

int32_t lin_scale(const float* in, float* out, const float* scale, const float* shift, const float* power, int mode,
                  long outer, long C, long inner, hipStream_t s) {
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(outer * C * inner)), dim3(kThreads), 0, s, in, out, scale, shift,
                       power, mode, outer, C, inner);
    return check_launch("lin_scale");
}

}  // namespace trtx
