// Round 6: which compute units a stream created with hipExtStreamCreateWithCUMask really gets on MI355X (256 CUs, 8 XCDs) - the bit -> (XCD, CU) map, found by
// launching a grid of busy workgroups on masked streams and reading HW_ID / XCC_ID in every workgroup.   hipcc --offload-arch=gfx950 -O2 -o bin/cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void where(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 0x7) << 8);
}
static int run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int n = 4096;
    unsigned* d;
    CHECK(hipMalloc(&d, n * 4));
    hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, s, d, 2000);
    CHECK(hipStreamSynchronize(s));
    std::vector<unsigned> h(n);
    CHECK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (unsigned v : h) per_xcc[v >> 16].insert(v & 0xffff);
    int bits = 0;
    for (uint32_t w : mask) bits += __builtin_popcount(w);
    size_t cus = 0;
    for (auto& kv : per_xcc) cus += kv.second.size();
    printf("%-34s bits set %3d -> %3zu CUs:", name, bits, cus);
    for (auto& kv : per_xcc) printf("  xcc%u:%zu", kv.first, kv.second.size());
    printf("\n");
    CHECK(hipFree(d));
    CHECK(hipStreamDestroy(s));
    return 0;
}
int main() {
    std::vector<uint32_t> m(8, 0);
    auto clear = [&]() { for (auto& w : m) w = 0; };
    for (auto& w : m) w = 0xffffffffu;
    run("all 256 bits", m);
    clear(); m[0] = 0xffffffffu; run("bits 0..31", m);
    clear(); m[0] = 0xffffffffu; m[1] = 0xffffffffu; run("bits 0..63", m);
    clear(); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); run("every 8th bit from 0", m);
    clear(); for (int i = 1; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); run("every 8th bit from 1", m);
    clear(); for (int i = 0; i < 256; i += 8) { m[i / 32] |= 1u << (i % 32); m[(i + 1) / 32] |= 1u << ((i + 1) % 32); } run("bits 8k and 8k+1", m);
    clear(); for (int i = 0; i < 256; i += 4) m[i / 32] |= 1u << (i % 32); run("every 4th bit", m);
    clear(); for (int i = 0; i < 256; i += 2) m[i / 32] |= 1u << (i % 32); run("every 2nd bit", m);
    clear(); m[7] = 0xffffffffu; run("bits 224..255", m);
    clear(); for (int i = 0; i < 128; ++i) m[i / 32] |= 1u << (i % 32); run("bits 0..127", m);
    return 0;
}
