// dev probe: shader-clock stamps inside the k-steps of conv_igemm_f16_kernel<4, 32> (64 -> 64, 3x3, HxH, batch 32)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ long long g_dbg[8][6];
#define TRTX_STAMP(i, kt) do { if (blockIdx.x == 7 && threadIdx.x == 0 && (kt) < 8) { g_dbg[kt][i] = clock64(); } } while (0)
#include "../../tensorrtx_amd/csrc/kernels/conv_igemm.hip"
using namespace trtx;
int main(int argc, char** argv) {
    const int N = 32, H = argc > 1 ? atoi(argv[1]) : 80, Cin = 64, Cout = 64, k = 3;
    ConvArgs a{};
    a.N = N; a.H = a.W = H; a.Cin = Cin; a.ld_in = Cin; a.Ho = a.Wo = H; a.Cout = Cout; a.Cout_pad = 64; a.ld_out = 64;
    a.kh = a.kw = k; a.stride_h = a.stride_w = 1; a.pad_h = a.pad_w = 1; a.dil_h = a.dil_w = 1; a.groups = 1;
    a.bk = 32; a.CinK = Cin; a.K = k * k * Cin; a.Kpad = a.K; a.M = N * H * H; a.act1 = ACT_SILU; a.bn = 64;
    void *in, *w, *out; float* bias;
    hipMalloc(&in, (size_t)a.M * Cin * 2); hipMalloc(&w, (size_t)64 * a.Kpad * 2); hipMalloc(&out, (size_t)a.M * 64 * 2); hipMalloc(&bias, 256);
    hipMemset(in, 0x11, (size_t)a.M * Cin * 2); hipMemset(w, 0x11, (size_t)64 * a.Kpad * 2); hipMemset(bias, 0, 256);
    a.in = in; a.wgt = w; a.out = out; a.bias = bias;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        conv_igemm_f16(a, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[8][6];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof h);
        printf("H=%d kernel %.1f us (blocks %d)\n", H, ms * 1e3, (a.M + 127) / 128);
        for (int kt = 0; kt < 8; ++kt)
            printf("  kt %d: p01 %5lld  p12 %5lld  p23 %5lld  p34 %5lld | step total %5lld cycles\n", kt, h[kt][1] - h[kt][0],
                   h[kt][2] - h[kt][1], h[kt][3] - h[kt][2], h[kt][4] - h[kt][3], kt ? h[kt][0] - h[kt - 1][0] : 0);
    }
    return 0;
}
