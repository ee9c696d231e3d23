// First-layer ("stem") convolution for gfx950: reads the network input as the caller hands it over —
// fp32 NCHW (LINEAR) with 1..4 channels — and writes NHWC fp16 with bias(+folded BN) and activation fused,
// so the layout/dtype conversion pass over the largest tensor of the network disappears.
// (YOLOv8 model.0: 3->16 3x3/2, yolov8/src/model.cpp:115; ResNet-50 conv1: 3->64 7x7/2, resnet50.cpp:165-170.)
//
// With K = kh*kw*Cin = 27..147 and Cout <= 64 this layer is far below the MFMA ridge: it is bound by the HBM
// read of the fp32 image and the fp16 store.  Two kernels:
//   * conv_stem_lds_kernel (Cout 16/32/64, K <= 160, W % 4 == 0): a workgroup stages the fp32 input patch of a 4x64
//     output tile in LDS with coalesced 16-byte LDS-DMA loads (borders range-checked to zero), then every lane
//     gathers the 8 taps of its (pixel, k-chunk) from LDS into the B fragment of v_mfma_f32_16x16x32_f16 (converted to
//     fp16 in registers); the weights sit in registers as A fragments, a wave stores 32..128 contiguous bytes per pixel.
//     Gathering the taps straight from global memory (per-lane 4-byte loads, stride-2 columns) was bound by the
//     L1 request rate: 120-140 us on the 3->16 3x3/2 640x640 batch-32 layer against a 44 us traffic floor.
//   * conv_stem_kernel (fallback, any Cout in 8..64): one thread per output pixel, fp32 FMAs, weights broadcast
//     from LDS.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float stem_act(float v, int act, float alpha) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case ACT_SILU: return v / (1.0f + __expf(-v));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        case ACT_MISH: return mish_ref(v);
        default: return v;
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void conv_stem_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // [K][COUT]
    const float* __restrict__ w = static_cast<const float*>(p.wgt);
    const int K = p.kh * p.kw * p.Cin;
    for (int i = threadIdx.x; i < K * COUT; i += 256) s_w[i] = w[i];
    __syncthreads();
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= p.M) return;
    const int wo = (int)(m % p.Wo);
    const long t = m / p.Wo;
    const int ho = (int)(t % p.Ho);
    const long n = t / p.Ho;
    const float* __restrict__ in = static_cast<const float*>(p.in) + n * p.Cin * p.H * p.W;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = p.bias ? p.bias[c] : 0.f;
    const int hi0 = ho * p.stride_h - p.pad_h, wi0 = wo * p.stride_w - p.pad_w;
    int tap = 0;
    for (int c = 0; c < p.Cin; ++c) {
        const float* plane = in + (long)c * p.H * p.W;
        for (int r = 0; r < p.kh; ++r) {
            const int hi = hi0 + r;
            const bool hok = (unsigned)hi < (unsigned)p.H;
            for (int q = 0; q < p.kw; ++q, ++tap) {
                const int wi = wi0 + q;
                float x = 0.f;
                if (hok && (unsigned)wi < (unsigned)p.W) x = plane[(long)hi * p.W + wi];
                const float4* wr = reinterpret_cast<const float4*>(s_w + tap * COUT);
#pragma unroll
                for (int c4 = 0; c4 < COUT / 4; ++c4) {
                    const float4 wv = wr[c4];
                    acc[c4 * 4 + 0] = fmaf(x, wv.x, acc[c4 * 4 + 0]);
                    acc[c4 * 4 + 1] = fmaf(x, wv.y, acc[c4 * 4 + 1]);
                    acc[c4 * 4 + 2] = fmaf(x, wv.z, acc[c4 * 4 + 2]);
                    acc[c4 * 4 + 3] = fmaf(x, wv.w, acc[c4 * 4 + 3]);
                }
            }
        }
    }
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out) + m * p.ld_out;
#pragma unroll
    for (int c8 = 0; c8 < COUT / 8; ++c8) {
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = round_to_half(stem_act(acc[c8 * 8 + e], p.act1, p.alpha1));
        *reinterpret_cast<half8*>(out + c8 * 8) = v;
    }
}

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kStemTW = 64, kStemTH = 4;  // output tile of a workgroup: 4 rows x 64 columns = 16 MFMA groups of 16 pixels

struct StemGeom {
    int PR, PCA;            // input patch rows / 4-float-aligned columns staged in LDS per channel
    int tiles_x, tiles_y;
    int chunks;             // 16-byte chunks of the patch (Cin * PR * PCA / 4)
};

// NFRAG = Cout / 16, KS = 32-wide slices of K (K = kh*kw*Cin <= 32*KS).
// Stage 1: the fp32 input patch of the tile (Cin x PR x PCA) goes HBM -> LDS with 16-byte LDS-DMA loads, rows and
//          columns outside the image are range-checked to zero by the buffer descriptor.
// Stage 2: every lane gathers the 8 taps of its (pixel, k-chunk) from LDS, converts to fp16 and feeds the B operand of
//          v_mfma_f32_16x16x32_f16; the weights stay in registers as A fragments.
// FRAMES: stage 1 does not copy an fp32 NCHW tensor but SAMPLES it: the network input is the letterboxed camera frame
// (yolov8/src/preprocess.cu:7-127: scale-about-the-centre warp-affine, bilinear, grey border, BGR -> RGB, / 255) and each patch pixel is
// computed from the uint8 HWC source image on the fly - the 4.9 MB fp32 image per 640 x 640 frame is never written or read.  The value
// of a pixel is formed by the reference's expression in the reference's order (see frame_pixel), so the stem sees the same floats as
// after the separate letterbox kernel (plugins/letterbox.hip) and the engine output is the same bits.
struct FramePack {
    StemFrame img[kStemMaxFrames];
};

// One letterboxed pixel (dx, dy) of frame `im`, all three network channels (R, G, B).  The map is a pure scale (d2s[1] = d2s[3] = +-0), so
// the source column depends on dx alone and the source row on dy alone: `cx` / `cy` carry what the reference computes per pixel -
// src_x = m_x1 * dx + m_y1 * dy + m_z1 + 0.5f evaluated left to right (m_y1 * dy = +-0 adds nothing), its floor and fraction.
struct AxisSample {
    float pos;   // src_x (or src_y)
    int low;     // floor
    float l;     // pos - low
};
__device__ __forceinline__ AxisSample frame_axis(float m, float z, int d) {
#pragma clang fp contract(off)
    AxisSample a;
    a.pos = m * (float)d + z + 0.5f;
    a.low = (int)floorf(a.pos);
    a.l = a.pos - (float)a.low;
    return a;
}
__device__ __forceinline__ void frame_pixel(const StemFrame& im, const AxisSample& cx, const AxisSample& cy, float& r, float& g, float& b) {
#pragma clang fp contract(off)
    float c0, c1, c2;   // B, G, R of the source
    if (cx.pos <= -1 || cx.pos >= im.w || cy.pos <= -1 || cy.pos >= im.h) {
        c0 = c1 = c2 = 128.0f;
    } else {
        const int x_low = cx.low, y_low = cy.low, x_high = x_low + 1, y_high = y_low + 1;
        const float lx = cx.l, ly = cy.l, hx = 1 - lx, hy = 1 - ly;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const uint8_t* base = static_cast<const uint8_t*>(im.src);
        const size_t line = (size_t)im.w * 3;
        float v[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = q < 2 ? y_low : y_high, xx = (q & 1) ? x_high : x_low;
            const bool in = (q < 2 ? y_low >= 0 : y_high < im.h) && ((q & 1) ? x_high < im.w : x_low >= 0);
            const uint8_t* px = base + (size_t)(in ? yy : 0) * line + (size_t)(in ? xx : 0) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[q][c] = in ? (float)px[c] : 128.0f;
        }
        c0 = w1 * v[0][0] + w2 * v[1][0] + w3 * v[2][0] + w4 * v[3][0];
        c1 = w1 * v[0][1] + w2 * v[1][1] + w3 * v[2][1] + w4 * v[3][1];
        c2 = w1 * v[0][2] + w2 * v[1][2] + w3 * v[2][2] + w4 * v[3][2];
    }
    r = c2 / 255.0f;
    g = c1 / 255.0f;
    b = c0 / 255.0f;
}

// what a workgroup keeps in registers for all of its tiles: the weights as MFMA A fragments, the patch offsets of its taps, its bias
template <int NFRAG, int KS>
struct StemRegs {
    half8 wf[NFRAG][KS];
    int l_off[KS][8];   // float index inside the patch of tap (c, r, q) relative to the pixel's top-left corner
    float bias4[NFRAG][4];
};
template <int NFRAG, int KS>
__device__ __forceinline__ void stem_setup(const ConvArgs& p, const StemGeom& g, StemRegs<NFRAG, KS>& R);
template <int NFRAG, int KS>
__device__ __forceinline__ void stem_compute(const ConvArgs& p, const StemGeom& g, const StemRegs<NFRAG, KS>& R, const float* s_patch, int n, int tx0, int ty0, int shift);
template <int NFRAG, int KS>
__device__ __forceinline__ void stem_stage2(const ConvArgs& p, const StemGeom& g, const float* s_patch, int n, int tx0, int ty0, int shift) {
    StemRegs<NFRAG, KS> R;
    stem_setup<NFRAG, KS>(p, g, R);
    stem_compute<NFRAG, KS>(p, g, R, s_patch, n, tx0, ty0, shift);
}

template <int NFRAG, int KS>
__global__ __launch_bounds__(256) void conv_stem_frames_kernel(const ConvArgs p, const StemGeom g, const FramePack fr) {
    extern __shared__ __attribute__((aligned(16))) float s_patch[];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx0 = (t % g.tiles_x) * kStemTW;
    t /= g.tiles_x;
    const int ty0 = (t % g.tiles_y) * kStemTH;
    const int n = t / g.tiles_y;
    const int hi_start = ty0 * p.stride_h - p.pad_h;
    const int wi_start = tx0 * p.stride_w - p.pad_w;
    const int al_start = (wi_start >= 0 ? wi_start : wi_start - 3) / 4 * 4;  // same patch geometry as the copying kernel
    const int shift = wi_start - al_start;
    const StemFrame& im = fr.img[n];
    // stage 1: every patch pixel = one letterboxed pixel (three planes), or 0 outside the network image (the convolution's padding)
    const int plane = g.PR * g.PCA;
    const float inv_pca = 1.0f / (float)g.PCA;
    for (int e = tid; e < plane; e += 256) {
        int pr = (int)((float)e * inv_pca);   // within +-1, fixed up exactly
        int pc = e - pr * g.PCA;
        if (pc < 0) { --pr; pc += g.PCA; }
        if (pc >= g.PCA) { ++pr; pc -= g.PCA; }
        const int hi = hi_start + pr, wi = al_start + pc;
        float r = 0.f, gg = 0.f, b = 0.f;
        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
            const AxisSample cx = frame_axis(im.d2s[0], im.d2s[2], wi), cy = frame_axis(im.d2s[4], im.d2s[5], hi);
            frame_pixel(im, cx, cy, r, gg, b);
        }
        s_patch[e] = r;
        s_patch[plane + e] = gg;
        s_patch[2 * plane + e] = b;
    }
    __syncthreads();
    stem_stage2<NFRAG, KS>(p, g, s_patch, n, tx0, ty0, shift);
}

// PERSISTENT (round 4): a workgroup sets its weights / tap tables up once and walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...; the grid is
// what fits on the chip at once (launch_lds).  RetinaFace's 1280 x 1280 batch 8 is 12 800 tiles: the 160-load set-up is paid 512 times, not 12 800.
template <int NFRAG, int KS>
__global__ __launch_bounds__(256) void conv_stem_lds_kernel(const ConvArgs p, const StemGeom g, unsigned in_bytes, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float s_patch[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    StemRegs<NFRAG, KS> R;
    stem_setup<NFRAG, KS>(p, g, R);
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    // tile coordinates
    int t = tile;
    const int tx0 = (t % g.tiles_x) * kStemTW;
    t /= g.tiles_x;
    const int ty0 = (t % g.tiles_y) * kStemTH;
    const int n = t / g.tiles_y;
    const int hi_start = ty0 * p.stride_h - p.pad_h;
    const int wi_start = tx0 * p.stride_w - p.pad_w;
    const int al_start = (wi_start >= 0 ? wi_start : wi_start - 3) / 4 * 4;  // floor to a multiple of 4
    const int shift = wi_start - al_start;
    // ---- stage 1
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, in_bytes, 0x00020000);
        const int cpr = g.PCA / 4;                  // chunks per patch row
        const float inv_cpr = 1.0f / (float)cpr, inv_pr = 1.0f / (float)g.PR;
        for (int base = 0; base < g.chunks; base += 256) {
            const int ci = base + tid;
            int row = (int)((float)ci * inv_cpr);   // (c * PR + pr); estimate within +-1, fixed up exactly
            int cq = ci - row * cpr;
            if (cq < 0) { --row; cq += cpr; }
            if (cq >= cpr) { ++row; cq -= cpr; }
            int c = (int)((float)row * inv_pr);
            int pr = row - c * g.PR;
            if (pr < 0) { --c; pr += g.PR; }
            if (pr >= g.PR) { ++c; pr -= g.PR; }
            const int hi = hi_start + pr, wi = al_start + cq * 4;
            const bool ok = ci < g.chunks && (unsigned)hi < (unsigned)p.H && wi >= 0 && wi + 3 < p.W;
            const unsigned off = ok ? (unsigned)(((((long)n * p.Cin + c) * p.H + hi) * p.W + wi) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(s_patch + (size_t)(base + wave * 64) * 4), 16, off, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.W & 3) {
        // Ragged width (Faster R-CNN's 1333): the 16-byte chunk that straddles the end of an image row was range-checked away whole
        // above; its 1..3 real pixels are fetched here, one patch row per thread (right-edge tiles only have any)
        const int cpr = g.PCA / 4, rows = p.Cin * g.PR, wtail = p.W & ~3;
        const int cq = (wtail - al_start) >> 2;
        if (cq >= 0 && cq < cpr && wtail >= al_start) {
            const float* __restrict__ src = static_cast<const float*>(p.in);
            for (int rrow = tid; rrow < rows; rrow += 256) {
                const int c = rrow / g.PR, pr = rrow - c * g.PR, hi = hi_start + pr;
                if ((unsigned)hi >= (unsigned)p.H) continue;
                const size_t base = (((size_t)n * p.Cin + c) * p.H + hi) * p.W;
                for (int e = 0; e < 4; ++e)
                    if (wtail + e < p.W) s_patch[((size_t)rrow * cpr + cq) * 4 + e] = src[base + wtail + e];
            }
        }
        __syncthreads();
    }
    stem_compute<NFRAG, KS>(p, g, R, s_patch, n, tx0, ty0, shift);
    __syncthreads();   // every wave is done with the patch before the next tile's DMA overwrites it
    }
}

// weights, tap tables and bias of a lane.  EVERY load is issued unconditionally from a clamped index and masked afterwards: written as
// `cond ? w[i] : 0` the compiler put each of the 160 loads of the 7x7 stem (176 with the bias) into its own exec-masked branch with an
// s_waitcnt vmcnt(0) behind it - 176 dependent round trips of ~700 cycles per workgroup, which is where conv_stem_lds_kernel<4, 5> spent its
// 148 us on ResNet-50's 224 x 224 batch 32 (12x its byte floor; profiles/r04_kernel_stats_c2_1ctx_lanes1.txt) and 888 us on RetinaFace's 1280 x 1280.
template <int NFRAG, int KS>
__device__ __forceinline__ void stem_setup(const ConvArgs& p, const StemGeom& g, StemRegs<NFRAG, KS>& R) {
    const int lane = threadIdx.x & 63;
    const int khw = p.kh * p.kw;
    const int K = khw * p.Cin;
    const float* __restrict__ w = static_cast<const float*>(p.wgt);  // [tap = (c*kh + r)*kw + q][Cout]
    const int kq = (lane >> 4) * 8;
    auto& wf = R.wf;
    auto& l_off = R.l_off;
    auto& bias4 = R.bias4;
    const float inv_khw = 1.0f / (float)khw, inv_kw = 1.0f / (float)p.kw;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ks * 32 + kq + e;
            // k < 160, khw <= 49: (k + 0.5) / d is never within rounding distance of an integer, the floor is exact
            const int c = (int)(((float)k + 0.5f) * inv_khw), rem = k - c * khw;
            const int r = (int)(((float)rem + 0.5f) * inv_kw), q = rem - r * p.kw;
            l_off[ks][e] = k < K ? (c * g.PR + r) * g.PCA + q : 0;  // padded taps: zero weights, any valid address
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) {
                const int co = j * 16 + (lane & 15);
                const float wv = w[(size_t)(k < K ? k : K - 1) * p.Cout + (co < p.Cout ? co : p.Cout - 1)];
                wf[j][ks][e] = (k < K && co < p.Cout) ? (_Float16)wv : (_Float16)0.f;
            }
        }
    const int ch4 = (lane >> 4) * 4;
    const float* __restrict__ bsrc = p.bias ? p.bias : w;   // (wave-uniform; without a bias the loaded values are masked away)
#pragma unroll
    for (int j = 0; j < NFRAG; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = j * 16 + ch4 + e;
            const float bv = bsrc[co < p.Cout ? co : p.Cout - 1];
            bias4[j][e] = (p.bias && co < p.Cout) ? bv : 0.f;
        }
}

// stage 2 of both stem kernels: the patch is in LDS, wave w owns tile row w (64 pixels = 4 groups)
template <int NFRAG, int KS>
__device__ __forceinline__ void stem_compute(const ConvArgs& p, const StemGeom& g, const StemRegs<NFRAG, KS>& R, const float* s_patch, int n, int tx0, int ty0,
                                             int shift) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ch4 = (lane >> 4) * 4;
    const auto& wf = R.wf;
    const auto& l_off = R.l_off;
    const auto& bias4 = R.bias4;
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
#pragma nounroll   // (unrolled, the <4, 5> instantiation is ~100 KB of code against a 64 KB instruction cache, walked once per tile)
    for (int gi = 0; gi < 4; ++gi) {
        const int ty = wave, tx = gi * 16 + (lane & 15);
        const int ho = ty0 + ty, wo = tx0 + tx;
        const float* src = s_patch + (ty * p.stride_h) * g.PCA + tx * p.stride_w + shift;
        floatx4 acc[NFRAG];
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 xf;
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[e] = (_Float16)src[l_off[ks][e]];
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][ks], xf, acc[j], 0, 0, 0);
        }
        if (ho >= p.Ho || wo >= p.Wo) continue;
        const long m = ((long)n * p.Ho + ho) * p.Wo + wo;
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const int co = j * 16 + ch4;
            if (co >= p.Cout) continue;
            half4_t o;
            if (p.act1 == ACT_SILU) {  // wave-uniform: pick the activation once, not per element
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[j][e] + bias4[j][e];
                    o[e] = round_to_half(x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)));
                }
            } else if (p.act1 == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[j][e] + bias4[j][e];
                    o[e] = (_Float16)(x > 0.f ? x : 0.f);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = round_to_half(stem_act(acc[j][e] + bias4[j][e], p.act1, p.alpha1));
            }
            *reinterpret_cast<half4_t*>(out + m * p.ld_out + co) = o;
        }
    }
}

template <int NFRAG, int KS>
void launch_lds(const ConvArgs& a, hipStream_t s) {
    StemGeom g;
    g.PR = (kStemTH - 1) * a.stride_h + a.kh;
    g.PCA = ((kStemTW - 1) * a.stride_w + a.kw + 3 + 3) / 4 * 4;  // + up to 3 floats of alignment slack
    g.tiles_x = (a.Wo + kStemTW - 1) / kStemTW;
    g.tiles_y = (a.Ho + kStemTH - 1) / kStemTH;
    g.chunks = a.Cin * g.PR * g.PCA / 4;
    const size_t lds = (size_t)((g.chunks + 255) / 256 * 256) * 16;  // whole 1 KiB DMA rows
    const unsigned in_bytes = (unsigned)((size_t)a.N * a.Cin * a.H * a.W * 4);
    const int total = a.N * g.tiles_x * g.tiles_y;
    // as many workgroups as are resident at once (registers and this LDS size decide), each walking total / grid tiles
    // (cached per thread, kernel instantiation, DEVICE and LDS size: a thread that moves to a GPU with another CU count must not reuse the figure - ADVICE r4)
    static thread_local int resident[3] = {-1, 0, 0};   // [0]: device, [1]: LDS bytes the figure was computed for, [2]: workgroups on the chip
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (resident[0] != dev || resident[1] != (int)lds || resident[2] <= 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_stem_lds_kernel<NFRAG, KS>, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        (void)hipGetLastError();
        resident[0] = dev;
        resident[1] = (int)lds;
        resident[2] = per_cu * cus;
    }
    const int grid = total < resident[2] ? total : resident[2];
    hipLaunchKernelGGL((conv_stem_lds_kernel<NFRAG, KS>), dim3((unsigned)grid), dim3(256), lds, s, a, g, in_bytes, total);
}

template <int NFRAG, int KS>
void launch_frames(const ConvArgs& a0, const StemFrame* frames, hipStream_t s) {
    StemGeom g;
    g.PR = (kStemTH - 1) * a0.stride_h + a0.kh;
    g.PCA = ((kStemTW - 1) * a0.stride_w + a0.kw + 3 + 3) / 4 * 4;
    g.tiles_x = (a0.Wo + kStemTW - 1) / kStemTW;
    g.tiles_y = (a0.Ho + kStemTH - 1) / kStemTH;
    g.chunks = a0.Cin * g.PR * g.PCA / 4;
    const size_t lds = (size_t)((g.chunks + 255) / 256 * 256) * 16;
    for (int n0 = 0; n0 < a0.N; n0 += kStemMaxFrames) {   // the frame descriptors travel in the kernel arguments, 64 images at a time
        ConvArgs a = a0;
        a.N = a0.N - n0 < kStemMaxFrames ? a0.N - n0 : kStemMaxFrames;
        a.M = a.N * a.Ho * a.Wo;
        a.out = static_cast<char*>(a0.out) + (size_t)n0 * a.Ho * a.Wo * a.ld_out * 2;
        FramePack fr;
        memset(&fr, 0, sizeof(fr));
        for (int i = 0; i < a.N; ++i) fr.img[i] = frames[n0 + i];
        hipLaunchKernelGGL((conv_stem_frames_kernel<NFRAG, KS>), dim3((unsigned)(a.N * g.tiles_x * g.tiles_y)), dim3(256), lds, s, a, g, fr);
    }
}

template <int COUT>
void launch(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)a.kh * a.kw * a.Cin * COUT * sizeof(float);
    hipLaunchKernelGGL(conv_stem_kernel<COUT>, dim3((unsigned)((a.M + 255) / 256)), dim3(256), lds, s, a);
}

}  // namespace

bool conv_stem_supported(const ConvArgs& a) {
    const bool cout_ok = a.Cout == 8 || a.Cout == 16 || a.Cout == 32 || a.Cout == 64;
    return cout_ok && a.Cin >= 1 && a.Cin <= 4 && a.groups == 1 && a.dil_h == 1 && a.dil_w == 1 && !a.residual &&
           a.act2 == ACT_NONE && (size_t)a.kh * a.kw * a.Cin * a.Cout * 4 <= 48 * 1024;
}

// in: fp32 NCHW [N][Cin][H][W]; wgt: fp32 [kh*kw*Cin (c,r,q order)][Cout]; out: NHWC fp16 (ld_out % 8 == 0, 16-B aligned)
int32_t conv_stem_nchw_f32(const ConvArgs& a, hipStream_t s) {
    if (!conv_stem_supported(a) || a.ld_out % 8 || (reinterpret_cast<uintptr_t>(a.out) & 15)) return TRTX_ERR_UNSUPPORTED;
    const int K = a.kh * a.kw * a.Cin;
    const size_t patch = (size_t)a.Cin * ((kStemTH - 1) * a.stride_h + a.kh) * ((kStemTW - 1) * a.stride_w + a.kw + 9) * 4;
    if (a.Cout % 16 == 0 && K <= 160 && patch <= 60 * 1024 &&
        (size_t)a.N * a.Cin * a.H * a.W * 4 < 2000000000u && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0) {
        const int ks = K <= 32 ? 1 : 5;
        bool done = true;
        if (a.Cout == 16 && ks == 1) launch_lds<1, 1>(a, s);
        else if (a.Cout == 32 && ks == 1) launch_lds<2, 1>(a, s);
        else if (a.Cout == 64 && ks == 1) launch_lds<4, 1>(a, s);
        else if (a.Cout == 16 && ks == 5) launch_lds<1, 5>(a, s);
        else if (a.Cout == 32 && ks == 5) launch_lds<2, 5>(a, s);
        else if (a.Cout == 64 && ks == 5) launch_lds<4, 5>(a, s);
        else done = false;
        if (done) return check_launch("conv_stem_nchw_f32");
    }
    switch (a.Cout) {
        case 8: launch<8>(a, s); break;
        case 16: launch<16>(a, s); break;
        case 32: launch<32>(a, s); break;
        case 64: launch<64>(a, s); break;
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return check_launch("conv_stem_nchw_f32");
}

int32_t conv_stem_frames_f32(const ConvArgs& a, const StemFrame* frames, hipStream_t s) {
    if (!frames || !conv_stem_supported(a) || a.ld_out % 8 || (reinterpret_cast<uintptr_t>(a.out) & 15) || a.Cin != 3) return TRTX_ERR_UNSUPPORTED;
    const int K = a.kh * a.kw * a.Cin;
    const size_t patch = (size_t)a.Cin * ((kStemTH - 1) * a.stride_h + a.kh) * ((kStemTW - 1) * a.stride_w + a.kw + 9) * 4;
    if (a.Cout % 16 || K > 160 || patch > 60 * 1024) return TRTX_ERR_UNSUPPORTED;
    for (int i = 0; i < a.N; ++i)
        if (!frames[i].src || frames[i].w < 1 || frames[i].h < 1) return TRTX_ERR_INVALID;
    const int ks = K <= 32 ? 1 : 5;
    if (a.Cout == 16 && ks == 1) launch_frames<1, 1>(a, frames, s);
    else if (a.Cout == 32 && ks == 1) launch_frames<2, 1>(a, frames, s);
    else if (a.Cout == 64 && ks == 1) launch_frames<4, 1>(a, frames, s);
    else if (a.Cout == 16 && ks == 5) launch_frames<1, 5>(a, frames, s);
    else if (a.Cout == 32 && ks == 5) launch_frames<2, 5>(a, frames, s);
    else if (a.Cout == 64 && ks == 5) launch_frames<4, 5>(a, frames, s);
    else return TRTX_ERR_UNSUPPORTED;
    return check_launch("conv_stem_frames_f32");
}

}  // namespace trtx
