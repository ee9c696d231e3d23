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
#define cudaMalloc hipMalloc
#define cudaFree hipFree
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
