// ORACLE — test infrastructure only (never part of the product, never included by tensorrtx_amd/).
// Spelling bridge that lets the UNMODIFIED reference plugin sources under /root/reference (*.cu written against the
// CUDA runtime) be compiled by hipcc as *user plugins* against this repo's include/NvInfer.h, so that the reference's own
// kernels run on the MI355X as the parity oracle (oracle/_ref/libref_plugins.so, see oracle/ref_build.py).
// Only the runtime names those few files use are bridged.
#pragma once
#include <hip/hip_runtime.h>

#define CUDA_VERSION 11080
#define CUDART_VERSION 11080

typedef hipError_t cudaError_t;
typedef hipStream_t cudaStream_t;
typedef hipEvent_t cudaEvent_t;
typedef hipDeviceProp_t cudaDeviceProp;
#define cudaSuccess hipSuccess
#ifdef REF_COMPAT_HOST_MALLOC_FALLBACK
// Builder pins that run on machines WITHOUT a GPU (oracle/ref_build.py BUILDERS with this define): the reference's calibrator allocates
// its device input buffer in its constructor (retinaface/calibrator.cpp:20) and aborts if that fails, although a build that finds a
// calibration cache never touches the buffer.  Without a device the allocation comes from the host heap.
#include <cstdlib>
static inline int ref_compat_have_device() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}
static inline hipError_t ref_compat_malloc(void** p, size_t bytes) {
    if (ref_compat_have_device()) return hipMalloc(p, bytes);
    *p = malloc(bytes);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t ref_compat_free(void* p) {
    if (ref_compat_have_device()) return hipFree(p);
    free(p);
    return hipSuccess;
}
template <typename T>
static inline hipError_t ref_compat_malloc(T** p, size_t bytes) { return ref_compat_malloc(reinterpret_cast<void**>(p), bytes); }
#define cudaMalloc ref_compat_malloc
#define cudaFree ref_compat_free
#else
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#endif
#define cudaMallocHost hipHostMalloc
#define cudaFreeHost hipHostFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyAsync hipMemcpyAsync
#define cudaMemset hipMemset
#define cudaMemsetAsync hipMemsetAsync
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
#define cudaStreamCreate hipStreamCreate
#define cudaStreamDestroy hipStreamDestroy
#define cudaStreamSynchronize hipStreamSynchronize
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
#define cudaSetDevice hipSetDevice
#define cudaGetDevice hipGetDevice
#define cudaGetDeviceProperties hipGetDeviceProperties
