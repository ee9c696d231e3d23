// dev probe (round 5): where a workgroup of the fp32 MFMA convolution (kernels/conv_igemm_f32.hip) spends its cycles IN STEADY STATE - the kernel sits at 0.60 of
// the fp32 MFMA peak even with 16 balanced tiles per CU and with its loads range-checked away, while a bare loop of the same instructions reaches 0.82
// (tools/hip/mfma_f32_rate.hip).  Shader-clock stamps (thread 0 of four workgroups from the MIDDLE of the grid) at entry, first tile issued, every k-step
// (wait / barrier / issue / compute), k-loop done, epilogue done.  Includes the product kernel source with the stamp macros defined.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Itensorrtx_amd/csrc tools/hip/igemm_f32_anatomy.hip -o tools/hip/bin/igemm_f32_anatomy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int kProbeWGs = 4, kSteps = 40;
__device__ long long g_mark[kProbeWGs][4];
__device__ long long g_step[kProbeWGs][kSteps][5];
__device__ int g_grid;
#define PROBE_SLOT() (((int)blockIdx.x >= g_grid / 2 && (int)blockIdx.x < g_grid / 2 + 32 && ((int)blockIdx.x - g_grid / 2) % 8 == 0) ? ((int)blockIdx.x - g_grid / 2) / 8 : -1)
#define TRTX_MARK(i) do { const int s__ = PROBE_SLOT(); if (s__ >= 0 && threadIdx.x == 0) g_mark[s__][i] = clock64(); } while (0)
#define TRTX_STAMP(i, kt) do { const int s__ = PROBE_SLOT(); if (s__ >= 0 && threadIdx.x == 0 && (kt) < kSteps) g_step[s__][kt][i] = clock64(); } while (0)
#include "../../tensorrtx_amd/csrc/kernels/conv_igemm_f32.hip"
using namespace trtx;
namespace trtx { LaunchProbe* conv_launch_probe() { return nullptr; } }

static void run(const char* name, int N, int H, int Cin, int Cout, int k, int bn, int bm) {
    ConvArgs a{};
    a.f32 = 1;
    a.N = N; a.H = a.W = H; a.Cin = Cin; a.ld_in = Cin; a.Ho = a.Wo = H; a.Cout = Cout; a.ld_out = Cout;
    a.kh = a.kw = k; a.stride_h = a.stride_w = 1; a.pad_h = a.pad_w = k / 2; a.dil_h = a.dil_w = 1; a.groups = 1;
    a.bk = 16; a.CinK = conv_igemm_f32_pick_cink(Cin); a.K = k * k * a.CinK; a.Kpad = (a.K + 15) / 16 * 16; a.M = N * H * H; a.act1 = ACT_SILU;
    a.Cout_pad = (Cout + 15) / 16 * 16; a.bn = bn; a.bm = bm;
    void *in, *w, *out; float* bias;
    hipMalloc(&in, (size_t)a.M * Cin * 4); hipMalloc(&w, (size_t)a.Cout_pad * a.Kpad * 4); hipMalloc(&out, (size_t)a.M * Cout * 4); hipMalloc(&bias, a.Cout_pad * 4);
    hipMemset(in, 0x11, (size_t)a.M * Cin * 4); hipMemset(w, 0x11, (size_t)a.Cout_pad * a.Kpad * 4); hipMemset(bias, 0, a.Cout_pad * 4);
    a.in = in; a.wgt = w; a.out = out; a.bias = bias;
    const int tiles = ((a.M + bm - 1) / bm) * (a.Cout_pad / bn), grid = (tiles + 7) / 8 * 8;
    hipMemcpyToSymbol(HIP_SYMBOL(g_grid), &grid, 4);
    void* flush; hipMalloc(&flush, 512u << 20);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(flush, rep, 512u << 20);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int32_t st0 = conv_igemm_f32(a, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        if (st0) { printf("launch failed %d\n", st0); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep < 2) continue;
        static long long m[kProbeWGs][4], st[kProbeWGs][kSteps][5];
        hipMemcpyFromSymbol(m, HIP_SYMBOL(g_mark), sizeof m);
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_step), sizeof st);
        const int nk = a.Kpad / 16;
        printf("%s: N %d %dx%d %d->%d k%d tile %dx%d: %d tiles, %d k-steps (%d cycles of MFMA each per wave); event interval %.1f us\n", name, N, H, H, Cin, Cout, k, bm, bn, tiles, nk,
               (bm / 64) * (bn / 16) * 4 * 32, ms * 1e3);
        for (int s = 0; s < kProbeWGs; ++s) {
            printf("  workgroup grid/2+%d: set-up -> first tile issued %5lld | k-loop %6lld | epilogue %5lld | total %6lld cycles\n", 8 * s, m[s][1] - m[s][0], m[s][2] - m[s][1],
                   m[s][3] - m[s][2], m[s][3] - m[s][0]);
            long long sw = 0, sb = 0, si = 0, sc = 0;
            const int n = nk < kSteps ? nk : kSteps;
            for (int kt = 0; kt < n; ++kt) { sw += st[s][kt][1] - st[s][kt][0]; sb += st[s][kt][2] - st[s][kt][1]; si += st[s][kt][3] - st[s][kt][2]; sc += st[s][kt][4] - st[s][kt][3]; }
            printf("     mean per k-step over %d steps: wait %lld  barrier %lld  issue %lld  compute %lld  | step to step %lld\n", n, sw / n, sb / n, si / n, sc / n,
                   n > 1 ? (st[s][n - 1][0] - st[s][0][0]) / (n - 1) : 0);
            printf("     steps 8..15 (wait, barrier, issue, compute):");
            for (int kt = 8; kt < n && kt < 16; ++kt) printf(" [%lld %lld %lld %lld]", st[s][kt][1] - st[s][kt][0], st[s][kt][2] - st[s][kt][1], st[s][kt][3] - st[s][kt][2], st[s][kt][4] - st[s][kt][3]);
            printf("\n");
        }
    }
    hipFree(in); hipFree(w); hipFree(out); hipFree(bias); hipFree(flush);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("# igemm_f32_anatomy on %s, clock64() stamps\n", prop.gcnArchName);
    run("64->64 3x3 @80", 32, 80, 64, 64, 3, 64, 128);
    run("64->64 3x3 @80", 32, 80, 64, 64, 3, 64, 64);
    run("64->64 3x3 @64 b128 (16 tiles/CU)", 128, 64, 64, 64, 3, 64, 128);
    run("128->128 1x1 @40", 32, 40, 128, 128, 1, 128, 128);
    run("384->128 1x1 @40", 32, 40, 384, 128, 1, 64, 128);
    return 0;
}
