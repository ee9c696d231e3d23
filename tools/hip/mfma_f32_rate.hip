// dev harness (round 5): what bounds the fp32 MFMA convolution (kernels/conv_igemm_f32.hip) at 0.56-0.60 of 157 TFLOP/s whatever its tile shape - and, with
// its loads range-checked away, still only 0.64?  Measures on the whole chip, for W workgroups of 4 waves per CU:
//   A  a bare loop of v_mfma_f32_16x16x4_f32 over 8 independent accumulators                       -> the rate the pipe really delivers, and at which CLOCK
//   B  A + per 32 MFMAs: 6 ds_read_b128 fragment reads                                             -> + LDS reads
//   C  B + s_barrier per 32 MFMAs                                                                  -> + the workgroup barrier
//   D  C + 3 buffer_load ... lds pieces per 32 MFMAs, range-checked away (no memory access)         -> + DMA issue
//   E  D with the loads real (an L2-resident 4 MB source)                                           -> + the fill
//   F  E with ROLES: 8 waves, waves 4-7 only fetch (issue, counted wait, barrier), waves 0-3 only multiply (barrier, fragment reads, MFMAs)
//   G  E with the three pieces pinned between the MFMAs (after the 4th, 14th, 24th) instead of wherever the compiler puts them
// clock64() counts shader cycles, wall_clock64() a constant 100 MHz: their ratio is the clock the kernel actually ran at.
// Build: hipcc --offload-arch=gfx950 -O3 tools/hip/mfma_f32_rate.hip -o tools/hip/bin/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ int g_random_data;   // 1: the LDS tiles hold pseudo-random floats in [-1, 1) instead of small integers (data-dependent power -> clock?)

template <int MODE>
__global__ __launch_bounds__(MODE == 5 ? 512 : 256) void k(const float* src, unsigned src_bytes, float* out, long long* clk, int steps) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 12288];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    floatx4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 a[2] = {floatx4{1.f, 2.f, 3.f, 4.f}, floatx4{.5f, .25f, .125f, 1.f}}, b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = floatx4{(float)lane, 1.f, 2.f, (float)j};
    if (MODE >= 1) {
        for (int i = threadIdx.x; i < 3 * 12288 / 4; i += 256) {
            unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            reinterpret_cast<float*>(smem)[i] = g_random_data ? ((float)(h & 0xffffff) / 8388608.0f - 1.0f) : (float)(i & 7);
        }
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    const long long c0 = clock64(), w0 = wall_clock64();
    unsigned off = (unsigned)((blockIdx.x * 256 + (threadIdx.x & 255)) * 16) % (src_bytes - 65536);
    if (MODE == 5 && wave >= 4) {   // the fetching role: nothing but issue, counted wait, barrier
        for (int s = 0; s < steps; ++s) {
            __builtin_amdgcn_s_barrier();
            const int nst = (s + 2) % 3;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + nst * 12288 + (4 * p + wave - 4) * 1024), 16, off + p * 16384, 0, 0, 0);
            off = (off + 49152) % (src_bytes - 65536);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    for (int s = 0; s < steps; ++s) {
        const int st = s % 3;
        if (MODE == 5) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 6) {
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (MODE >= 2 && MODE <= 4) {
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 3 || MODE == 4) {
            const int nst = (s + 2) % 3;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned voff = MODE == 3 ? 0x80000000u : off + p * 16384;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + nst * 12288 + (4 * p + wave) * 1024), 16, voff, 0, 0, 0);
            }
            off = (off + 49152) % (src_bytes - 65536);
        }
        if (MODE >= 1) {
            const char* sb = smem + st * 12288 + lane * 16;
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const floatx4*>(sb + (wave * 2 + i) * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const floatx4*>(sb + 8192 + j * 1024);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][s4], a[i][s4], acc[i * 4 + j], 0, 0, 0);
                    const int n = s4 * 8 + j * 2 + i;
                    if (MODE == 6 && (n == 3 || n == 13 || n == 23)) {
                        const int p = n / 10;
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + ((s + 2) % 3) * 12288 + (4 * p + wave) * 1024), 16, off + p * 16384, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        if (MODE == 6) off = (off + 49152) % (src_bytes - 65536);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = clock64(), w1 = wall_clock64();
    floatx4 t = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += acc[i];
    if (t[0] == 12345.f) out[threadIdx.x] = t[1] + t[2] + t[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

template <int MODE>
static void run(const char* what, int wg_per_cu, int steps, const float* src, unsigned src_bytes, float* out, long long* clk) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    float best = 1e30f;
    long long h[2] = {0, 0};
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(MODE == 5 ? 512 : 256), 0, 0, src, src_bytes, out, clk, steps);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        }
    }
    const double flop = (double)grid * 4 * steps * 32 * 2048.0;
    printf("%-44s %d wg/CU %6d steps: %8.1f us  %6.1f TFLOP/s (%.2f of 157.3)  shader clock %.0f MHz, %.0f cycles per 32-MFMA step\n", what, wg_per_cu, steps, best * 1e3,
           flop / best / 1e9, flop / best / 1e9 / 157.3, h[1] ? 100.0 * h[0] / h[1] : 0.0, (double)h[0] / steps);
}

int main() {
    const unsigned src_bytes = 4u << 20;
    float *src, *out;
    long long* clk;
    CHECK(hipMalloc(&src, src_bytes));
    CHECK(hipMemset(src, 0, src_bytes));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&clk, 64));
    for (int rnd : {1}) {
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_random_data), &rnd, 4));
    printf("## LDS operand tiles hold %s\n", rnd ? "pseudo-random floats in [-1, 1)" : "small integers (0..7)");
    for (int steps : {2000}) {
        for (int w : {1, 2, 3, 4}) {
            run<0>("A bare MFMA loop", w, steps, src, src_bytes, out, clk);
            run<1>("B + 6 ds_read_b128 per step", w, steps, src, src_bytes, out, clk);
            run<2>("C + s_barrier per step", w, steps, src, src_bytes, out, clk);
            run<3>("D + 3 LDS-DMA pieces per step, range-checked", w, steps, src, src_bytes, out, clk);
            run<4>("E + 3 LDS-DMA pieces per step, from L2", w, steps, src, src_bytes, out, clk);
            run<5>("F = E with roles (4 fetching + 4 multiplying waves)", w, steps, src, src_bytes, out, clk);
            run<6>("G = E, pieces pinned between the MFMAs", w, steps, src, src_bytes, out, clk);
        }
    }
    }
    return 0;
}
