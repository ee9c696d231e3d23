// dev probe (round 4): where the time of ONE small convolution launch goes - shader-clock stamps at the entry of a workgroup, after its first tile
// is issued, at every k-step (wait / barrier / issue / compute), after the k-loop and after the epilogue, for the layers of YOLOv8n that sit at
// 9-17 us against byte floors of 1.5-3 us.  Includes the product kernel source with the stamp macros defined.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Itensorrtx_amd/csrc tools/hip/igemm_launch_anatomy.hip -o tools/hip/bin/igemm_launch_anatomy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int kProbeWGs = 4;   // workgroups 0, 1, last - 1, last report (by dispatch index)
__device__ long long g_mark[kProbeWGs][4];
__device__ long long g_step[kProbeWGs][12][5];
__device__ int g_grid;
#define PROBE_SLOT() ((int)blockIdx.x < 2 ? (int)blockIdx.x : (((int)blockIdx.x >= g_grid - 2 && (int)blockIdx.x < g_grid) ? 2 + ((int)blockIdx.x - (g_grid - 2)) : -1))
#define TRTX_MARK(i) do { const int s__ = PROBE_SLOT(); if (s__ >= 0 && threadIdx.x == 0) g_mark[s__][i] = clock64(); } while (0)
#define TRTX_STAMP(i, kt) do { const int s__ = PROBE_SLOT(); if (s__ >= 0 && threadIdx.x == 0 && (kt) < 12) g_step[s__][kt][i] = clock64(); } while (0)
#include "../../tensorrtx_amd/csrc/kernels/conv_igemm.hip"
using namespace trtx;

// variant 0: the host dispatch's default (128-row tile, four waves); 1: its 64-row tile (four waves of 16 rows, twice the workgroups);
// 2: a 128-row tile on EIGHT waves of 16 rows (two waves per SIMD: half the epilogue items and DMA pieces per wave)
template <int NFRAG, bool ONE>
static void launch8(const ConvArgs& a) {
    const int tiles_n = a.Cout_pad / (16 * NFRAG), total = ((a.M + 127) / 128) * tiles_n, chunk = (total + 7) / 8;
    const unsigned in_bytes = (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ld_in + a.Cin) * 2), w_bytes = (unsigned)((size_t)a.Cout_pad * a.Kpad * 2);
    hipLaunchKernelGGL((conv_igemm_f16_kernel<NFRAG, 32, 1, false, 1, 1, 0, false, 8, false, false, ONE>), dim3(chunk * 8), dim3(512), 0, 0, a, in_bytes, w_bytes, tiles_n, total, chunk, 0);
}
// variants 3 / 4: the default 128-row four-wave tile with SIX / EIGHT LDS stages (five / seven tiles in flight instead of two)
template <int NFRAG, bool ONE, int NST>
static void launch_deep(const ConvArgs& a) {
    const int tiles_n = a.Cout_pad / (16 * NFRAG), total = ((a.M + 127) / 128) * tiles_n, chunk = (total + 7) / 8;
    const unsigned in_bytes = (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ld_in + a.Cin) * 2), w_bytes = (unsigned)((size_t)a.Cout_pad * a.Kpad * 2);
    hipLaunchKernelGGL((conv_igemm_f16_kernel<NFRAG, 32, 1, false, 2, 1, NST, false, 4, false, false, ONE>), dim3(chunk * 8), dim3(256), 0, 0, a, in_bytes, w_bytes, tiles_n, total, chunk, 0);
}
static void run(const char* name, int N, int H, int Cin, int Cout, int k, int bk, int variant = 0) {
    ConvArgs a{};
    a.N = N; a.H = a.W = H; a.Cin = Cin; a.ld_in = Cin; a.Ho = a.Wo = H; a.Cout = Cout; a.ld_out = Cout;
    a.kh = a.kw = k; a.stride_h = a.stride_w = 1; a.pad_h = a.pad_w = k / 2; a.dil_h = a.dil_w = 1; a.groups = 1;
    a.bk = bk; a.CinK = Cin; a.K = k * k * Cin; a.Kpad = a.K; a.M = N * H * H; a.act1 = ACT_SILU;
    a.bn = conv_igemm_pick_bn(Cout); a.Cout_pad = (Cout + a.bn - 1) / a.bn * a.bn; a.t_wsk = 1; a.t_ws = 1;
    void *in, *w, *out; float* bias;
    hipMalloc(&in, (size_t)a.M * Cin * 2); hipMalloc(&w, (size_t)a.Cout_pad * a.Kpad * 2); hipMalloc(&out, (size_t)a.M * Cout * 2); hipMalloc(&bias, a.Cout_pad * 4);
    hipMemset(in, 0x11, (size_t)a.M * Cin * 2); hipMemset(w, 0x11, (size_t)a.Cout_pad * a.Kpad * 2); hipMemset(bias, 0, a.Cout_pad * 4);
    a.in = in; a.wgt = w; a.out = out; a.bias = bias;
    const int rows = variant == 1 ? 64 : 128;
    const int tiles = ((a.M + rows - 1) / rows) * (a.Cout_pad / a.bn), grid = (tiles + 7) / 8 * 8;
    hipMemcpyToSymbol(HIP_SYMBOL(g_grid), &grid, 4);
    void* flush; hipMalloc(&flush, 512u << 20);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(flush, rep, 512u << 20);   // evict: the layer's input comes from memory, as after its producer's launch
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (variant == 2) {
            if (a.bn == 128) { if (k == 1) launch8<8, true>(a); else launch8<8, false>(a); }
            else { if (k == 1) launch8<4, true>(a); else launch8<4, false>(a); }
        } else if (variant == 3) {
            if (a.bn == 128) { if (k == 1) launch_deep<8, true, 6>(a); else launch_deep<8, false, 6>(a); }
            else { if (k == 1) launch_deep<4, true, 6>(a); else launch_deep<4, false, 6>(a); }
        } else if (variant == 4) {
            if (a.bn == 128) { if (k == 1) launch_deep<8, true, 8>(a); else launch_deep<8, false, 8>(a); }
            else { if (k == 1) launch_deep<4, true, 8>(a); else launch_deep<4, false, 8>(a); }
        } else {
            a.bm = variant == 1 ? 64 : 0;
            conv_igemm_f16(a, 0);
        }
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep < 2) continue;
        long long m[kProbeWGs][4], st[kProbeWGs][12][5];
        hipMemcpyFromSymbol(m, HIP_SYMBOL(g_mark), sizeof m);
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_step), sizeof st);
        const int nk = a.Kpad / bk;
        printf("[variant %d] %s: N %d %dx%d %d->%d k%d bk %d: M %d, %d tiles, %d k-steps; event interval around the launch %.1f us\n", variant, name, N, H, H, Cin, Cout, k, bk, a.M, tiles, nk, ms * 1e3);
        long long first = m[0][0];
        for (int s = 0; s < kProbeWGs; ++s) first = m[s][0] < first ? m[s][0] : first;
        for (int s = 0; s < kProbeWGs; ++s) {
            printf("  workgroup %s: entry at +%6lld | set-up -> first tile issued %5lld | -> k-loop done %6lld | epilogue %5lld | total %6lld cycles\n",
                   s == 0 ? "0     " : s == 1 ? "1     " : s == 2 ? "last-1" : "last  ", m[s][0] - first, m[s][1] - m[s][0], m[s][2] - m[s][1], m[s][3] - m[s][2], m[s][3] - m[s][0]);
            printf("     k-steps (wait, barrier, issue, compute):");
            for (int kt = 0; kt < nk && kt < 8; ++kt) printf(" [%lld %lld %lld %lld]", st[s][kt][1] - st[s][kt][0], st[s][kt][2] - st[s][kt][1], st[s][kt][3] - st[s][kt][2], st[s][kt][4] - st[s][kt][3]);
            printf("\n");
        }
    }
    hipFree(in); hipFree(w); hipFree(out); hipFree(bias); hipFree(flush);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("# igemm_launch_anatomy on %s, clock64() stamps (shader clock, %d kHz reported)\n", prop.gcnArchName, prop.clockRate);
    for (int v = 0; v < 5; ++v) {
        run("model.9.cv1-like", 32, 20, 256, 128, 1, 32, v);
        run("C2f m.cv1 @20", 32, 20, 128, 128, 3, 32, v);
        run("C2f m.cv1 @40", 32, 40, 64, 64, 3, 32, v);
        run("1x1 64->64 @20", 32, 20, 64, 64, 1, 32, v);
        run("64->64 3x3 @80", 32, 80, 64, 64, 3, 32, v);
        run("1x1 128->128 @40", 32, 40, 128, 128, 1, 32, v);
    }
    return 0;
}
