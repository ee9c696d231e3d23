// dev probe (round 6): issue cost of plain / packed / transcendental VALU instructions for 1, 2, 3 waves per SIMD - shader-clock cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/hip/valu_rate.hip -o tools/hip/bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ long long g_cyc[8];
template <int KIND>
__global__ void k(float* out, int iters, int slot) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, 0x3f7fff00, %0" : "+v"(v[i]));
            if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            if (KIND == 3) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));
        }
        if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&v[i]));
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) g_cyc[slot] = t1 - t0;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const char* names[5] = {"v_mul_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_f16_f32", "v_pk_mul_f32 (8 per 16 values)"};
    for (int kind = 0; kind < 5; ++kind)
        for (int wps = 1; wps <= 4; ++wps) {
            const int iters = 200;
            void (*fn)(float*, int, int) = kind == 0 ? k<0> : kind == 1 ? k<1> : kind == 2 ? k<2> : kind == 3 ? k<3> : k<4>;
            hipLaunchKernelGGL(fn, dim3(256), dim3(256 * wps), 0, 0, out, iters, 0);
            hipLaunchKernelGGL(fn, dim3(256), dim3(256 * wps), 0, 0, out, iters, 0);
            hipDeviceSynchronize();
            long long c; hipMemcpyFromSymbol(&c, HIP_SYMBOL(g_cyc), 8);
            const int n = kind == 4 ? 8 : 16;
            printf("%-32s %d wave(s)/SIMD: %6.2f cycles per instruction and wave, %6.2f per instruction and SIMD\n", names[kind], wps, (double)c / (iters * n), (double)c / (iters * n * wps));
        }
    return 0;
}
