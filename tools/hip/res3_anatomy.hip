// dev probe (round 6): phase anatomy of the resident-operand 3x3 kernel (conv_res.hip) - shader-clock stamps of the first wave of each half at every phase
// (entry, k-loop done / patch DMA issued, epilogue done, waits retired, past the barrier) for a few workgroups, plus the launch's event time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Itensorrtx_amd/csrc tools/hip/res3_anatomy.hip -o tools/hip/bin/res3_anatomy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int kWG = 4, kPh = 12;
__device__ long long g_st[kWG][3][kPh][8];
__device__ long long g_t0[kWG];
#define PROBE_SLOT() ((int)blockIdx.x == 0 ? 0 : (int)blockIdx.x == 9 ? 1 : (int)blockIdx.x == 100 ? 2 : (int)blockIdx.x == 201 ? 3 : -1)
#define TRTX_RES_STAMP(ph, i) do { const int s__ = PROBE_SLOT(); if (s__ >= 0 && (threadIdx.x & 255) == 0 && (ph) < kPh) g_st[s__][threadIdx.x >> 8][ph][i] = clock64(); } while (0)
__device__ int g_ablate;
#define TRTX_RES_ABLATE g_ablate
#include "../../tensorrtx_amd/csrc/kernels/conv_res.hip"
namespace trtx { LaunchProbe* conv_launch_probe() { return nullptr; } const Options& options() { static Options o; return o; } }
using namespace trtx;

static void run(const char* name, int N, int H, int Cin, int Cout, bool res, bool flushed = true, int ablate = 0) {
    hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &ablate, 4);
    ConvArgs a{};
    a.N = N; a.H = a.W = H; a.Cin = Cin; a.ld_in = Cin; a.Ho = a.Wo = H; a.Cout = Cout; a.ld_out = Cout; a.ld_res = Cout;
    a.kh = a.kw = 3; a.stride_h = a.stride_w = 1; a.pad_h = a.pad_w = 1; a.dil_h = a.dil_w = 1; a.groups = 1;
    a.bk = 32; a.CinK = (Cin + 31) / 32 * 32; a.K = 9 * a.CinK; a.Kpad = a.K; a.M = N * H * H; a.act1 = ACT_SILU;
    a.bn = Cout; a.Cout_pad = Cout; a.t_wsk = 1; a.t_ws = 7;
    void *in, *w, *out, *rs; float* bias;
    hipMalloc(&in, (size_t)a.M * Cin * 2); hipMalloc(&w, (size_t)a.Cout_pad * a.Kpad * 2); hipMalloc(&out, (size_t)a.M * Cout * 2); hipMalloc(&rs, (size_t)a.M * Cout * 2); hipMalloc(&bias, a.Cout_pad * 4);
    hipMemset(in, 0x11, (size_t)a.M * Cin * 2); hipMemset(w, 0x11, (size_t)a.Cout_pad * a.Kpad * 2); hipMemset(bias, 0, a.Cout_pad * 4); hipMemset(rs, 0x11, (size_t)a.M * Cout * 2);
    a.in = in; a.wgt = w; a.out = out; a.bias = bias; a.residual = res ? rs : nullptr;
    if (!conv_res_possible(a)) { printf("%s: not possible\n", name); return; }
    void* flush; hipMalloc(&flush, 512u << 20);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        if (flushed) hipMemset(flush, rep, 512u << 20);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        conv_res_f16(&a, 1, 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
    }
    static long long st[kWG][3][kPh][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    if (ablate) printf("[ablate %d: %s%s%s%s] ", ablate, ablate & 1 ? "no stores " : "", ablate & 2 ? "no patch DMA " : "", ablate & 4 ? "no k-loop " : "", ablate & 8 ? "no activation" : "");
    printf("%s%s: event interval %.1f us (best of 3, input %s)\n", name, res ? " + shortcut" : "", best * 1e3f, flushed ? "flushed" : "warm: the previous run's");
    for (int s = 0; s < kWG; ++s) {
        const long long t0 = st[s][0][0][0];
        printf("  workgroup probe %d (cycles since its first stamp; per phase: half: entry | role done | epilogue done | waited | past barrier)\n", s);
        for (int ph = 0; ph < kPh; ++ph) {
            if (!st[s][0][ph][4] && !st[s][1][ph][4] && !st[s][2][ph][4]) break;
            printf("    ph %2d", ph);
            for (int h = 0; h < 3; ++h) {
                printf("   h%d:", h);
                if (!st[s][h][ph][0]) continue;
                for (int i = 0; i < 5; ++i) printf(" %6lld", st[s][h][ph][i] ? st[s][h][ph][i] - t0 : -1);
            }
            printf("\n");
        }
    }
    static long long z[kWG][3][kPh][8];
    hipMemcpyToSymbol(HIP_SYMBOL(g_st), z, sizeof(z));
    hipFree(in); hipFree(w); hipFree(out); hipFree(rs); hipFree(bias); hipFree(flush);
}
// fp32 operands (ConvArgs::f32, conv_igemm_f32.hip's layouts): column tile `bn`
static void run_f32(const char* name, int N, int H, int Cin, int Cout, int bn) {
    int ablate = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &ablate, 4);
    ConvArgs a{};
    a.f32 = 1;
    a.N = N; a.H = a.W = H; a.Cin = Cin; a.ld_in = Cin; a.Ho = a.Wo = H; a.Cout = Cout; a.ld_out = Cout; a.ld_res = Cout;
    a.kh = a.kw = 3; a.stride_h = a.stride_w = 1; a.pad_h = a.pad_w = 1; a.dil_h = a.dil_w = 1; a.groups = 1;
    a.bk = 16; a.CinK = (Cin + 15) / 16 * 16; a.K = 9 * a.CinK; a.Kpad = a.K; a.M = N * H * H; a.act1 = ACT_SILU;
    a.bn = bn; a.bm = 128; a.Cout_pad = (Cout + 15) / 16 * 16; a.t_ws = 7;
    void *in, *w, *out; float* bias;
    hipMalloc(&in, (size_t)a.M * Cin * 4); hipMalloc(&w, (size_t)a.Cout_pad * a.Kpad * 4); hipMalloc(&out, (size_t)a.M * Cout * 4); hipMalloc(&bias, a.Cout_pad * 4);
    hipMemset(in, 0x3c, (size_t)a.M * Cin * 4); hipMemset(w, 0x3c, (size_t)a.Cout_pad * a.Kpad * 4); hipMemset(bias, 0, a.Cout_pad * 4);
    a.in = in; a.wgt = w; a.out = out; a.bias = bias;
    if (!conv_res_possible(a)) { printf("%s: not possible\n", name); return; }
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        conv_res_f16(&a, 1, 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
    }
    static long long st[kWG][3][kPh][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    const double gf = 2.0 * a.M * 9 * Cin * Cout / 1e9;
    printf("[fp32] %s, column tile %d: event interval %.1f us = %.2f of the fp32 MFMA peak\n", name, bn, best * 1e3f, gf / (best * 1e-3) / 1e3 / 157.3);
    for (int s = 0; s < 2; ++s) {
        const long long t0 = st[s][0][0][0];
        printf("  workgroup probe %d (cycles since its first stamp; per phase and group: entry | role done | epilogue done | waited | past barrier)\n", s);
        for (int ph = 0; ph < 6; ++ph) {
            printf("    ph %2d", ph);
            for (int h = 0; h < 2; ++h) {
                printf("   g%d:", h);
                for (int i = 0; i < 5; ++i) printf(" %6lld", st[s][h][ph][i] ? st[s][h][ph][i] - t0 : -1);
            }
            printf("\n");
        }
    }
    static long long z[kWG][3][kPh][8];
    hipMemcpyToSymbol(HIP_SYMBOL(g_st), z, sizeof(z));
    hipFree(in); hipFree(w); hipFree(out); hipFree(bias);
}
int main(int argc, char** argv) {
    if (argc > 1) {
        run_f32("64->64 3x3 @80 b32", 32, 80, 64, 64, 32);
        run_f32("64->64 3x3 @80 b32", 32, 80, 64, 64, 16);
        run_f32("32->32 3x3 @80 b32", 32, 80, 32, 32, 32);
        return 0;
    }
    run("64->64 3x3 @80 b32", 32, 80, 64, 64, false);
    run("64->64 3x3 @80 b32", 32, 80, 64, 64, false, false);
    for (int ab : {4, 7, 15}) run("64->64 3x3 @80 b32", 32, 80, 64, 64, false, false, ab);
    run("64->64 3x3 @80 b32", 32, 80, 64, 64, true);
    run("64->64 3x3 @40 b32", 32, 40, 64, 64, false);
    run("32->32 3x3 @80 b32", 32, 80, 32, 32, false);
    run("64->80 3x3 @80 b32", 32, 80, 64, 80, false);
    return 0;
}
