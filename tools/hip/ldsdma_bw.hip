// dev probe: per-CU throughput of the operand paths the conv kernels can use, on L2-resident data:
//   mode 0  buffer_load_dwordx4 ... lds   (LDS-DMA, 1 KiB per wave-instruction)
//   mode 1  global_load_dwordx4 -> VGPR
//   mode 2  LDS-DMA + ds_read_b128 of what landed (the igemm2 operand path without MFMA)
// Every wave streams through a window shared by the whole chip (default 2 MiB: misses L1, hits L2), keeping DEPTH
// wave-instructions in flight.  Prints B/clk/CU at 2.4 GHz for 1..4 workgroups (4..16 waves) per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, unsigned window, int iters, float* sink, unsigned stride) {
    __shared__ __attribute__((aligned(16))) char smem[4 * DEPTH * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, window, 0x00020000);
    unsigned off = (unsigned)(((size_t)blockIdx.x * 4 + wave) * 40960u) % window;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    char* my = smem + wave * DEPTH * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            unsigned o = (off + d * 1024u + lane * 16u) % window;
            if (MODE == 3)  // conv A-operand pattern: 16 rows x 64 B per instruction, rows `stride` bytes apart
                o = (off + (unsigned)(d * 16 + (lane >> 2)) * stride + (lane & 3) * 16u + (unsigned)(it & 1) * 64u) % window;
            if (MODE == 1) {
                const floatx4 v = *reinterpret_cast<const floatx4*>(src + o);
                acc += v;
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(my + d * 1024), 16, o, 0, 0, 0);
            }
        }
        if (MODE == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += *reinterpret_cast<const floatx4*>(my + d * 1024 + lane * 16);
        }
        off = (off + (MODE == 3 ? ((it & 1) ? DEPTH * 16u * stride : 0u) : DEPTH * 1024u)) % window;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int MODE, int DEPTH>
void run(const char* name, const char* src, unsigned window, float* sink, unsigned stride = 0) {
    for (int wgs = 1; wgs <= 4; ++wgs) {
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(256 * wgs), dim3(256), 0, 0, src, window, iters, sink, stride);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double bytes = 256.0 * wgs * 4 * iters * DEPTH * 1024.0;
        printf("%-28s depth %d  %2d waves/CU: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", name, DEPTH, wgs * 4, best * 1e3,
               bytes / best / 1e9, bytes / (best * 1e-3) / 2.4e9 / 256.0);
    }
}

int main(int argc, char** argv) {
    const unsigned window = (argc > 1 ? atoi(argv[1]) : 2048) * 1024u;
    char* src;
    float* sink;
    hipMalloc(&src, window + 4096);
    hipMalloc(&sink, 256);
    hipMemset(src, 0, window + 4096);
    printf("window %u KiB\n", window / 1024);
    run<0, 3>("LDS-DMA", src, window, sink);
    run<0, 6>("LDS-DMA", src, window, sink);
    run<1, 3>("global_load->VGPR", src, window, sink);
    run<1, 6>("global_load->VGPR", src, window, sink);
    run<3, 3>("LDS-DMA rows 64B/128B", src, window, sink, 128);
    run<3, 3>("LDS-DMA rows 64B/256B", src, window, sink, 256);
    run<3, 3>("LDS-DMA rows 64B/64B", src, window, sink, 64);
    run<2, 3>("LDS-DMA + ds_read", src, window, sink);
    run<2, 6>("LDS-DMA + ds_read", src, window, sink);
    return 0;
}
