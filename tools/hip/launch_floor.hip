// dev harness (round 4): what a kernel boundary costs on MI355X against a device-wide barrier inside one persistent kernel - the number
// that decides whether a persistent multi-layer kernel for the small YOLOv8n layers can pay.
//   A  empty kernel, G workgroups of 256 threads, back to back on one stream                      -> launch-to-launch time
//   B  kernel whose workgroups each write 16 KB (a 20x20 / 40x40 layer's output is 3-26 MB)       -> + the end-of-kernel write-back
//   C  ONE persistent kernel (one workgroup per CU x R): L rounds of (write 16 KB per workgroup; device-wide barrier with release / acquire
//      fences at agent scope, atomic counter in global memory)                                     -> time per round
// Build: hipcc --offload-arch=gfx950 -O3 tools/hip/launch_floor.hip -o tools/hip/bin/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1024) *p = 1; }

__global__ __launch_bounds__(256) void write_kernel(float4* out, int round) {
    float4 v = make_float4((float)round, 1.f, 2.f, 3.f);
    float4* dst = out + (size_t)blockIdx.x * 1024;   // 16 KB per workgroup
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i * 256 + threadIdx.x] = v;
}

__global__ __launch_bounds__(256) void persistent_kernel(float4* out, unsigned* counter, int rounds, int chunks_per_wg) {
    const unsigned nwg = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        for (int c = 0; c < chunks_per_wg; ++c) {
            float4 v = make_float4((float)r, 1.f, 2.f, 3.f);
            float4* dst = out + ((size_t)c * nwg + blockIdx.x) * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i * 256 + threadIdx.x] = v;
        }
        // device-wide barrier: release our writes, arrive, spin until everyone of this round arrived, acquire
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);   // (agent scope through the builtin below)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned target = (unsigned)(r + 1) * nwg;
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;   // (bail out instead of hanging the box should the workgroups ever not be co-resident)
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 20000000) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

static float timed(void (*body)(void*), void* ctx, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        body(ctx);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best * 1e3f;
}

struct Ctx { int grid, launches, chunks; float4* out; unsigned* counter; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# launch_floor on %s (%d CUs): us per kernel boundary vs us per device-wide barrier inside one persistent kernel; best of 5\n", prop.gcnArchName, cus);
    float4* out;
    unsigned* counter;
    CHECK(hipMalloc(&out, (size_t)8192 * 16384));
    CHECK(hipMalloc(&counter, 4));
    const int L = 200;
    for (int grid : {256, 512, 2048, 8192}) {
        Ctx c{grid, L, 1, out, counter};
        const float a = timed([](void* p) { Ctx* c = (Ctx*)p; for (int i = 0; i < c->launches; ++i) hipLaunchKernelGGL(empty_kernel, dim3(c->grid), dim3(256), 0, 0, (int*)nullptr); }, &c, 5);
        const float b = timed([](void* p) { Ctx* c = (Ctx*)p; for (int i = 0; i < c->launches; ++i) hipLaunchKernelGGL(write_kernel, dim3(c->grid), dim3(256), 0, 0, c->out, i); }, &c, 5);
        printf("grid %5d workgroups: empty kernel %6.2f us / launch   16 KB-per-workgroup writer (%5.1f MB) %6.2f us / launch\n", grid, a / L, grid * 16384 / 1e6, b / L);
    }
    for (int per_cu : {1, 2}) {
        for (int chunks : {1, 4, 16}) {
            Ctx c{cus * per_cu, L, chunks, out, counter};
            const float t = timed([](void* p) {
                Ctx* c = (Ctx*)p;
                CHECK(hipMemsetAsync(c->counter, 0, 4, 0));
                hipLaunchKernelGGL(persistent_kernel, dim3(c->grid), dim3(256), 0, 0, c->out, c->counter, c->launches, c->chunks);
            }, &c, 5);
            printf("persistent, %4d workgroups, %2d x 16 KB each per round (%5.1f MB): %6.2f us / round (write + device-wide barrier)\n", c.grid, chunks,
                   (double)c.grid * chunks * 16384 / 1e6, t / L);
        }
    }
    return 0;
}
