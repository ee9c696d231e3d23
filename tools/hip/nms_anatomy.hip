// dev probe (round 6): where the fused NMS kernel's time goes - shader-clock stamps of wave 0 / wave 10 of image 0 at the phase boundaries
// (0 entry, 1 keys built, 2 sorted, 3 records in LDS, 4 in-block words done, 5 block loop done, 6 end) on a synthetic decode buffer of ~700 candidates per image.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Itensorrtx_amd/csrc -Itensorrtx_amd/csrc/plugins tools/hip/nms_anatomy.hip -o tools/hip/bin/nms_anatomy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__device__ long long g_st[2][8];
#define TRTX_NMS_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 640)) g_st[threadIdx.x ? 1 : 0][i] = clock64(); } while (0)
#include "experiments/yolo_nms_fused.hip"

int main(int argc, char** argv) {
    const int batch = 32, max_out = 1000, det = 90, ncls = argc > 1 ? atoi(argv[1]) : 1, n = argc > 2 ? atoi(argv[2]) : 700;
    std::vector<float> h((size_t)batch * (1 + max_out * det), 0.f);
    srand(1);
    for (int b = 0; b < batch; ++b) {
        float* img = h.data() + (size_t)b * (1 + max_out * det);
        img[0] = (float)n;
        for (int i = 0; i < n; ++i) {
            float* d = img + 1 + (size_t)i * det;
            const float cx = rand() % 600 + 20, cy = rand() % 600 + 20, w = rand() % 60 + 10, hh = rand() % 60 + 10;
            d[0] = cx - w / 2; d[1] = cy - hh / 2; d[2] = cx + w / 2; d[3] = cy + hh / 2;
            d[4] = 0.11f + (rand() % 1000) * 0.0008f; d[5] = (float)(rand() % ncls);
        }
    }
    float* dec; int *ki, *kc; float* kd; void* ws;
    hipMalloc(&dec, h.size() * 4); hipMemcpy(dec, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&ki, batch * max_out * 4); hipMalloc(&kc, batch * 4); hipMalloc(&kd, (size_t)batch * max_out * 6 * 4);
    const size_t wsb = trtx_yolo_nms_workspace(batch); hipMalloc(&ws, wsb);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        trtx_yolo_nms(dec, batch, max_out, 0.1f, 0.45f, ki, kc, kd, ws, wsb, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep) best = ms < best ? ms : best;
    }
    long long st[2][8]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    int cnt[32]; hipMemcpy(cnt, kc, sizeof(cnt), hipMemcpyDeviceToHost);
    printf("%d candidates per image, %d classes: event interval %.1f us, kept %d on image 0\n", n, ncls, best * 1e3f, cnt[0]);
    for (int w = 0; w < 2; ++w) {
        printf("  wave %2d:", w ? 10 : 0);
        for (int i = 1; i < 7; ++i) printf("  %s %6lld", i == 1 ? "keys" : i == 2 ? "sort" : i == 3 ? "records" : i == 4 ? "in-block" : i == 5 ? "block loop" : "emit", st[w][i] - st[w][i - 1]);
        printf("   (cycles)\n");
    }
    return 0;
}
