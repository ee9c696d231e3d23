// dev probe: shader clock (s_memtime) vs constant 100 MHz wall clock, for short and long kernels
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long* out, int iters) {
    long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0001f, 0.5f);
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main() {
    long long* d; hipMalloc(&d, 64);
    long long h[3];
    for (int rep = 0; rep < 3; ++rep)
    for (int iters : {1000, 10000, 100000, 1000000}) {
        for (int blocks : {1, 256 * 8}) {
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("iters %8d blocks %5d: shader cycles %10lld wall ticks %8lld -> %.0f MHz, %.2f cycles/iter\n", iters, blocks, h[0], h[1],
                   h[1] ? 100.0 * h[0] / h[1] : 0.0, (double)h[0] / iters);
        }
    }
    return 0;
}
