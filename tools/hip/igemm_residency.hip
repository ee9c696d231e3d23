// dev probe (round 5): WHERE and WHEN every workgroup of the fp16 implicit-GEMM / resident-patch convolution runs (the fp32 twin: igemm_f32_residency.hip).
// Every workgroup records the 100 MHz wall clock at entry and exit and its CU; printed per launch: tiles per CU, co-resident workgroups, CU-time with no workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Itensorrtx_amd/csrc tools/hip/igemm_residency.hip tensorrtx_amd/csrc/kernels/conv_ws.hip \
//         tensorrtx_amd/csrc/kernels/conv_gemm256.hip tensorrtx_amd/csrc/options.cpp -o tools/hip/bin/igemm_residency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>
constexpr int kMaxWG = 1 << 15;
__device__ long long g_rec[kMaxWG][2];
__device__ unsigned g_cu[kMaxWG];
__device__ __forceinline__ unsigned cu_key() {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 0xf) << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 0x7) << 8);   // cu_id[11:8] + sh_id[12] | se_id[15:13] | xcc
}
#define TRTX_MARK(i) do { if (threadIdx.x == 0 && (int)blockIdx.x < kMaxWG) { if ((i) == 0) { g_rec[blockIdx.x][0] = wall_clock64(); g_cu[blockIdx.x] = cu_key(); } if ((i) == 3) g_rec[blockIdx.x][1] = wall_clock64(); } } while (0)
#include "../../tensorrtx_amd/csrc/kernels/conv_igemm.hip"
using namespace trtx;

static void run(const char* name, int N, int H, int Cin, int Cout, int k, int bn, int bm, int ws = 1) {
    ConvArgs a{};
    a.N = N; a.H = a.W = H; a.Cin = Cin; a.ld_in = Cin; a.Ho = a.Wo = H; a.Cout = Cout; a.ld_out = Cout;
    a.kh = a.kw = k; a.stride_h = a.stride_w = 1; a.pad_h = a.pad_w = k / 2; a.dil_h = a.dil_w = 1; a.groups = 1;
    a.bk = 32; a.CinK = conv_igemm_pick_cink(Cin, 32); a.K = k * k * a.CinK; a.Kpad = (a.K + 31) / 32 * 32; a.M = N * H * H; a.act1 = ACT_SILU;
    a.Cout_pad = (Cout + 15) / 16 * 16; a.bn = bn; a.bm = bm; a.t_ws = ws; a.t_wsk = 1;
    void *in, *w, *out; float* bias;
    hipMalloc(&in, (size_t)a.M * Cin * 2); hipMalloc(&w, (size_t)a.Cout_pad * a.Kpad * 2); hipMalloc(&out, (size_t)a.M * Cout * 2); hipMalloc(&bias, a.Cout_pad * 4);
    hipMemset(in, 0x11, (size_t)a.M * Cin * 2); hipMemset(w, 0x11, (size_t)a.Cout_pad * a.Kpad * 2); hipMemset(bias, 0, a.Cout_pad * 4);
    a.in = in; a.wgt = w; a.out = out; a.bias = bias;
    const int tiles = ((a.M + bm - 1) / bm) * (a.Cout_pad / bn), grid = (tiles + 7) / 8 * 8;
    const double gflop = 2.0 * a.M * k * k * Cin * Cout / 1e9;
    void* flush; hipMalloc(&flush, 512u << 20);
    static long long rec[kMaxWG][2];
    static unsigned cu[kMaxWG];
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(flush, rep, 512u << 20);
        { void* d; hipGetSymbolAddress(&d, HIP_SYMBOL(g_rec)); hipMemset(d, 0, sizeof(long long) * 2 * kMaxWG); }
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int32_t st0 = conv_igemm_f16(a, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        if (st0) { printf("%s: launch failed %d\n", name, st0); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep < 2) continue;
        hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_rec), sizeof(long long) * 2 * std::min(grid, kMaxWG));
        hipMemcpyFromSymbol(cu, HIP_SYMBOL(g_cu), sizeof(unsigned) * std::min(grid, kMaxWG));
        long long t0 = 1LL << 62, t1 = 0;
        std::map<unsigned, std::vector<std::pair<long long, long long>>> per;
        int n = 0;
        for (int b = 0; b < grid && b < kMaxWG; ++b) {
            if (!rec[b][1]) continue;   // (workgroups beyond the last tile return before the first mark... they record nothing)
            ++n;
            t0 = std::min(t0, rec[b][0]); t1 = std::max(t1, rec[b][1]);
            per[cu[b]].push_back({rec[b][0], rec[b][1]});
        }
        const double span = (double)(t1 - t0);   // 10 ns ticks
        double busy_sum = 0, resident_sum = 0, first_start_max = 0, last_end_min = 1e30, wg_life = 0;
        int tmin = 1 << 30, tmax = 0;
        for (auto& kv : per) {
            auto& v = kv.second;
            tmin = std::min<int>(tmin, v.size()); tmax = std::max<int>(tmax, v.size());
            std::vector<std::pair<long long, int>> ev;
            long long fs = 1LL << 62, le = 0;
            for (auto& iv : v) { ev.push_back({iv.first, 1}); ev.push_back({iv.second, -1}); fs = std::min(fs, iv.first); le = std::max(le, iv.second); resident_sum += iv.second - iv.first; wg_life += iv.second - iv.first; }
            std::sort(ev.begin(), ev.end());
            int c = 0; long long prev = 0;
            for (auto& e : ev) { if (c > 0) busy_sum += e.first - prev; c += e.second; prev = e.first; }
            first_start_max = std::max(first_start_max, (double)(fs - t0));
            last_end_min = std::min(last_end_min, (double)(le - t0));
        }
        const int ncu = (int)per.size();
        printf("%s: N %d %dx%d %d->%d k%d tile %dx%d ws %d: %d tiles on %d CUs, event interval %.1f us = %.2f of the algorithmic HBM time at 8 TB/s (in + out + weights once); first entry -> last exit %.1f us\n", name, N, H, H, Cin, Cout,
               k, bm, bn, ws, n, ncu, ms * 1e3, ((double)a.M * (Cin + Cout) * 2 + (double)a.Cout_pad * a.Kpad * 2) / 8e12 / (ms * 1e-3), span * 0.01);
        printf("    tiles per CU min %d mean %.2f max %d | a workgroup lives %.1f us on average | co-resident workgroups on a busy CU: %.2f | CU-time with no workgroup: %.1f %% of %d CUs x span"
               " (of 256 x span: %.1f %%) | latest first entry +%.1f us, earliest last exit -%.1f us\n", tmin, (double)n / ncu, tmax, wg_life / n * 0.01, resident_sum / busy_sum,
               100.0 * (1.0 - busy_sum / (ncu * span)), ncu, 100.0 * (1.0 - busy_sum / (256 * span)), first_start_max * 0.01, (span - last_end_min) * 0.01);
    }
    hipFree(in); hipFree(w); hipFree(out); hipFree(bias); hipFree(flush);
}

int main() {
    run("64->64 3x3 @80 b32", 32, 80, 64, 64, 3, 64, 128);
    run("64->64 3x3 @80 b32 patch", 32, 80, 64, 64, 3, 64, 128, 3);
    run("64->64 3x3 @80 b32", 32, 80, 64, 64, 3, 64, 64);
    run("128->128 3x3 @40 b32", 32, 40, 128, 128, 3, 128, 128);
    run("128->128 3x3 @40 b32 patch", 32, 40, 128, 128, 3, 128, 128, 3);
    run("64->64 3x3 @40 b32", 32, 40, 64, 64, 3, 64, 128);
    run("128->128 3x3 @20 b32", 32, 20, 128, 128, 3, 64, 64);
    run("384->128 1x1 @40 b32", 32, 40, 384, 128, 1, 128, 128);
    run("128->64 1x1 @80 b32", 32, 80, 128, 64, 1, 64, 128);
    run("64->64 3x3 @64 b64 (8 tiles/CU)", 64, 64, 64, 64, 3, 64, 128);
    run("64->64 3x3 @64 b8 (1 tile/CU)", 8, 64, 64, 64, 3, 64, 128);
    return 0;
}
