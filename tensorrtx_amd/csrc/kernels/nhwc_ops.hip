// NHWC activation-tensor kernels for gfx950: layout/dtype conversion, pooling, nearest resize,
// element-wise, activation, per-channel scale, channel-slice copy, spatial mean, and the generic
// direct (de)convolution used for shapes the MFMA implicit-GEMM kernel does not cover.
//
// These are the engine-side implementations of the TensorRT layers the reference builders add
// around the convolutions: addPoolingNd (yolov8/src/block.cpp:219-233 SPPF, resnet/resnet50.cpp:172),
// addResize NEAREST (yolov8/src/model.cpp:145-158), addElementWise (block.cpp:106, resnet50.cpp:146),
// addActivation, addScale, addConcatenation fall-backs, addReduce AVG (rcnn/rcnn.cpp:166).
//
// All are HBM-bound: one pass over the data, 16-byte accesses per lane whenever the channel count and
// the channel stride allow it (VEC = 8 halfs / 4 floats), grid-stride loops capped at 256 CUs x 8 WGs.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../common.h"
#include "kernels.h"

namespace trtx {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256 * 8;

inline int grid_for(long work) {
    long b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    return (int)(b > kMaxBlocks ? kMaxBlocks : b);
}

template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
    T v[V];
};

template <typename T>
__device__ __forceinline__ float to_f(T x) {
    return (float)x;
}
template <typename T>
__device__ __forceinline__ T from_f(float x) {
    return (T)x;
}

__device__ __forceinline__ float act_f(float v, int act, float alpha) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_TANH: return tanhf(v);
        case ACT_MISH: return mish_ref(v);
        default: return v;
    }
}

__device__ __forceinline__ float ew_f(float a, float b, int op) {
    switch (op) {
        case EW_SUM: return a + b;
        case EW_PROD: return a * b;
        case EW_MAX: return a > b ? a : b;
        case EW_MIN: return a < b ? a : b;
        case EW_SUB: return a - b;
        case EW_DIV: return a / b;
        case EW_POW: return powf(a, b);
        default: return a;
    }
}

// ---- layout conversion -------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int C, int H, int W,
                                    int Cpad, int ld) {
    const long HW = (long)H * W;
    const long total = (long)N * HW * Cpad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long px = i / Cpad;  // n*HW + hw
        const long n = px / HW;
        const long hw = px - n * HW;
        const float v = c < C ? in[(n * C + c) * HW + hw] : 0.f;
        out[px * ld + c] = from_f<T>(v);
    }
}

// fp32 NCHW -> fp16 NHWC as a tiled transpose through LDS: a block moves 64 channels x 32 pixels, reading 128-byte runs
// along HW and writing 16-byte channel chunks (128 bytes per pixel row of the tile).  The element-wise kernel above reads
// with a stride of HW floats per lane: 0.5 TB/s on the (1000, 1024, 14, 14) RoIAlign tensor of the R-CNN graph.
__global__ __launch_bounds__(256) void nchw_to_nhwc_f16_tiled_kernel(const float* __restrict__ in, _Float16* __restrict__ out,
                                                                     int C, long HW, int Cpad, int ld) {
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    __shared__ float tile[64][33];
    const long n = blockIdx.z;
    const long hw0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 pixels x 8 channel rows per pass
#pragma unroll
    for (int r = ty; r < 64; r += 8) {
        const int c = c0 + r;
        const long hw = hw0 + tx;
        tile[r][tx] = (c < C && hw < HW) ? in[(n * C + c) * HW + hw] : 0.f;  // channels [C, Cpad) are written as zeros
    }
    __syncthreads();
    const int px = threadIdx.x >> 3, c8 = threadIdx.x & 7;  // 32 pixels x 8 chunks of 8 channels
    const long hw = hw0 + px;
    const int c = c0 + c8 * 8;
    if (hw < HW && c < Cpad) {
        half8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)tile[c8 * 8 + e][px];
        *reinterpret_cast<half8_t*>(out + (n * HW + hw) * ld + c) = v;
    }
}

// tiled transpose through LDS: reads coalesced along C (NHWC), writes coalesced along HW (NCHW)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int C, long HW, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long hw0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const long hw = hw0 + r;
        const int c = c0 + tx;
        float v = 0.f;
        if (hw < HW && c < C) v = to_f(in[((long)n * HW + hw) * ld + c]);
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long hw = hw0 + tx;
        if (hw < HW && c < C) out[((long)n * C + c) * HW + hw] = tile[tx][r];
    }
}

// ---- pooling --------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void pool_kernel(const T* __restrict__ in, T* __restrict__ out, int op, int N, int H, int W, int C,
                            int ld_in, int Ho, int Wo, int ld_out, int kh, int kw, int sh, int sw, int ph, int pw,
                            int avg_excl) {
    const int CV = C / V;
    const long total = (long)N * Ho * Wo * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long px = i / CV;
        const int wo = (int)(px % Wo);
        px /= Wo;
        const int ho = (int)(px % Ho);
        const long n = px / Ho;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = op == POOL_MAX ? -INFINITY : 0.f;
        int cnt = 0;
        for (int r = 0; r < kh; ++r) {
            const int hi = ho * sh - ph + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < kw; ++q) {
                const int wi = wo * sw - pw + q;
                if ((unsigned)wi >= (unsigned)W) continue;
                const Pack<T, V> x = *reinterpret_cast<const Pack<T, V>*>(in + ((n * H + hi) * W + wi) * ld_in + cv * V);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float f = to_f(x.v[e]);
                    acc[e] = op == POOL_MAX ? (f > acc[e] ? f : acc[e]) : acc[e] + f;
                }
                ++cnt;
            }
        }
        Pack<T, V> o;
        const float div = avg_excl ? (float)(cnt > 0 ? cnt : 1) : (float)(kh * kw);
#pragma unroll
        for (int e = 0; e < V; ++e) o.v[e] = from_f<T>(op == POOL_MAX ? acc[e] : acc[e] / div);
        *reinterpret_cast<Pack<T, V>*>(out + ((n * Ho + ho) * Wo + wo) * ld_out + cv * V) = o;
    }
}

// SPPF: three chained k x k stride-1 'same' max-pools of one small map.  A workgroup owns (image, 8-channel chunk): the whole
// H x W map of that chunk sits in LDS (two ping-pong planes), every stage reads k*k neighbours from LDS and writes its
// output both to global memory (a channel slice of the SPPF concat buffer) and to the other plane.
// T / V: _Float16 x 8 (fp16 engines) or float x 4 (round 6: fp32 engines ran the three pools as three launches) - a 16-byte chunk of channels either way.
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool_chain3_kernel(const T* __restrict__ in, T* __restrict__ o1, T* __restrict__ o2, T* __restrict__ o3, int H,
                                                             int W, int C, int ld_in, int ld1, int ld2, int ld3, int k) {
    typedef T half8_t __attribute__((ext_vector_type(V)));
    extern __shared__ __attribute__((aligned(16))) char s_map_raw[];  // [2][(H + 2r) * (W + 2r)], -inf border: no bound tests
    half8_t* s_map = reinterpret_cast<half8_t*>(s_map_raw);
    const int HW = H * W;
    const int chunks = C / V;
    const int cv = blockIdx.x % chunks;
    const long n = blockIdx.x / chunks;
    const int r = k / 2;
    const int PW = W + 2 * r, PHW = (H + 2 * r) * PW;
    half8_t vinf;
#pragma unroll
    for (int e = 0; e < V; ++e) vinf[e] = (T)(-__builtin_inff());
    for (int p = threadIdx.x; p < 2 * PHW; p += 256) s_map[p] = vinf;
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
        const int h = p / W, w = p - h * W;
        s_map[(h + r) * PW + w + r] = *reinterpret_cast<const half8_t*>(in + (n * HW + p) * ld_in + cv * V);
    }
    __syncthreads();
    T* outs[3] = {o1, o2, o3};
    const int lds[3] = {ld1, ld2, ld3};
#pragma unroll
    for (int stage = 0; stage < 3; ++stage) {
        const half8_t* src = s_map + (stage & 1) * PHW;
        half8_t* dst = s_map + ((stage + 1) & 1) * PHW;
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int h = p / W, w = p - h * W;
            const half8_t* win = src + h * PW + w;  // top-left corner of the k x k window in padded coordinates
            half8_t m = win[0];
            for (int dy = 0; dy < k; ++dy)
                for (int dx = 0; dx < k; ++dx) m = __builtin_elementwise_max(m, win[dy * PW + dx]);  // v_pk_max_f16 / v_max_f32
            dst[(h + r) * PW + w + r] = m;
            *reinterpret_cast<half8_t*>(outs[stage] + (n * HW + p) * lds[stage] + cv * V) = m;
        }
        __syncthreads();
    }
}

template <typename T, int V>
__global__ void resize_nearest_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int C,
                                      int ld_in, int Ho, int Wo, int ld_out) {
    const int CV = C / V;
    const long total = (long)N * Ho * Wo * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long px = i / CV;
        const int wo = (int)(px % Wo);
        px /= Wo;
        const int ho = (int)(px % Ho);
        const long n = px / Ho;
        int hi = (int)(((long)ho * H) / Ho);
        int wi = (int)(((long)wo * W) / Wo);
        hi = hi < H ? hi : H - 1;
        wi = wi < W ? wi : W - 1;
        *reinterpret_cast<Pack<T, V>*>(out + ((n * Ho + ho) * Wo + wo) * ld_out + cv * V) =
                *reinterpret_cast<const Pack<T, V>*>(in + ((n * H + hi) * W + wi) * ld_in + cv * V);
    }
}

template <typename T, int V>
__global__ void elementwise_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int op,
                                   long pixels, int C, int ld_a, int ld_b, int ld_out) {
    const int CV = C / V;
    const long total = pixels * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long px = i / CV;
        const Pack<T, V> x = *reinterpret_cast<const Pack<T, V>*>(a + px * ld_a + cv * V);
        const Pack<T, V> y = *reinterpret_cast<const Pack<T, V>*>(b + px * ld_b + cv * V);
        Pack<T, V> o;
#pragma unroll
        for (int e = 0; e < V; ++e) o.v[e] = from_f<T>(ew_f(to_f(x.v[e]), to_f(y.v[e]), op));
        *reinterpret_cast<Pack<T, V>*>(out + px * ld_out + cv * V) = o;
    }
}

template <typename T, int V>
__global__ void activation_kernel(const T* __restrict__ in, T* __restrict__ out, int act, float alpha, long pixels,
                                  int C, int ld_in, int ld_out) {
    const int CV = C / V;
    const long total = pixels * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long px = i / CV;
        const Pack<T, V> x = *reinterpret_cast<const Pack<T, V>*>(in + px * ld_in + cv * V);
        Pack<T, V> o;
#pragma unroll
        for (int e = 0; e < V; ++e) o.v[e] = from_f<T>(act_f(to_f(x.v[e]), act, alpha));
        *reinterpret_cast<Pack<T, V>*>(out + px * ld_out + cv * V) = o;
    }
}

template <typename T, int V>
__global__ void scale_kernel(const T* __restrict__ in, T* __restrict__ out, const float* __restrict__ scale,
                             const float* __restrict__ shift, long pixels, int C, int ld_in, int ld_out) {
    const int CV = C / V;
    const long total = pixels * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long px = i / CV;
        const Pack<T, V> x = *reinterpret_cast<const Pack<T, V>*>(in + px * ld_in + cv * V);
        Pack<T, V> o;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int c = cv * V + e;
            const float sc = scale ? scale[c] : 1.f;
            const float sh = shift ? shift[c] : 0.f;
            o.v[e] = from_f<T>(to_f(x.v[e]) * sc + sh);
        }
        *reinterpret_cast<Pack<T, V>*>(out + px * ld_out + cv * V) = o;
    }
}

template <typename T, int V>
__global__ void copy_kernel(const T* __restrict__ in, T* __restrict__ out, long pixels, int C, int ld_in,
                            int ld_out) {
    const int CV = C / V;
    const long total = pixels * CV;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long px = i / CV;
        *reinterpret_cast<Pack<T, V>*>(out + px * ld_out + cv * V) =
                *reinterpret_cast<const Pack<T, V>*>(in + px * ld_in + cv * V);
    }
}

template <typename T>
__global__ void reduce_hw_avg_kernel(const T* __restrict__ in, T* __restrict__ out, int HW, int C, int ld_in,
                                     int ld_out) {
    // one workgroup per (n, 64-channel group); threads = 64 channels x 4 pixel lanes
    const int n = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
        for (int p = part; p < HW; p += 4) acc += to_f(in[((long)n * HW + p) * ld_in + c]);
    __shared__ float s[4][64];
    s[part][threadIdx.x & 63] = acc;
    __syncthreads();
    if (part == 0 && c < C) {
        const float t = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
        out[(long)n * ld_out + c] = from_f<T>(t / (float)HW);
    }
}

// fp16, C % 8 == 0: a lane owns 8 consecutive channels (16-byte loads instead of 2-byte ones: the element-wise kernel above took 654 us for
// the 803 MB of res5's output on 4 000 RoIs, a tenth of HBM speed).  Same partition of the pixels over four lanes and the same order of
// additions as above, so the averages are the same bits.
__global__ __launch_bounds__(256) void reduce_hw_avg_f16x8_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, int HW, int C, int ld_in,
                                                                  int ld_out) {
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    const int n = blockIdx.y;
    const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    const int part = threadIdx.x >> 6;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int p = part; p < HW; p += 4) {
            const half8_t v = *reinterpret_cast<const half8_t*>(in + ((long)n * HW + p) * ld_in + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    __shared__ float s[4][64][9];   // padded: the four parts of a lane land in distinct banks
#pragma unroll
    for (int e = 0; e < 8; ++e) s[part][threadIdx.x & 63][e] = acc[e];
    __syncthreads();
    if (part == 0 && c < C) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = s[0][threadIdx.x][e] + s[1][threadIdx.x][e] + s[2][threadIdx.x][e] + s[3][threadIdx.x][e];
            o[e] = from_f<_Float16>(t / (float)HW);
        }
        *reinterpret_cast<half8_t*>(out + (long)n * ld_out + c) = o;
    }
}

// ---- generic direct convolution / transposed convolution -----------------------------------------------
template <typename T>
__global__ void conv_direct_kernel(const ConvArgs p) {
    const T* __restrict__ in = static_cast<const T*>(p.in);
    const float* __restrict__ w = static_cast<const float*>(p.wgt);
    const T* __restrict__ res = static_cast<const T*>(p.residual);
    T* __restrict__ out = static_cast<T*>(p.out);
    const int cin_g = p.Cin / p.groups;
    const int cout_g = p.Cout / p.groups;
    const long total = (long)p.M * p.Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % p.Cout);
        const long m = i / p.Cout;
        const int wo = (int)(m % p.Wo);
        const long t = m / p.Wo;
        const int ho = (int)(t % p.Ho);
        const long n = t / p.Ho;
        const int g = co / cout_g;
        float acc = p.bias ? p.bias[co] : 0.f;
        const float* wk = w + (size_t)co * p.kh * p.kw * cin_g;
        for (int r = 0; r < p.kh; ++r) {
            const int hi = ho * p.stride_h - p.pad_h + r * p.dil_h;
            if ((unsigned)hi >= (unsigned)p.H) continue;
            for (int q = 0; q < p.kw; ++q) {
                const int wi = wo * p.stride_w - p.pad_w + q * p.dil_w;
                if ((unsigned)wi >= (unsigned)p.W) continue;
                const T* ip = in + ((n * p.H + hi) * p.W + wi) * p.ld_in + g * cin_g;
                const float* wp = wk + (r * p.kw + q) * cin_g;
                for (int c = 0; c < cin_g; ++c) acc = fmaf(to_f(ip[c]), wp[c], acc);
            }
        }
        acc = act_f(acc, p.act1, p.alpha1);
        if (res) acc += to_f(res[m * p.ld_res + co]);
        acc = act_f(acc, p.act2, p.alpha2);
        out[m * p.ld_out + co] = from_f<T>(acc);
    }
}

// weights pre-arranged as [Cout][kh][kw][Cin/groups] (gather form)
template <typename T>
__global__ void deconv_direct_kernel(const ConvArgs p) {
    const T* __restrict__ in = static_cast<const T*>(p.in);
    const float* __restrict__ w = static_cast<const float*>(p.wgt);
    T* __restrict__ out = static_cast<T*>(p.out);
    const int cin_g = p.Cin / p.groups;
    const int cout_g = p.Cout / p.groups;
    const long total = (long)p.M * p.Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % p.Cout);
        const long m = i / p.Cout;
        const int wo = (int)(m % p.Wo);
        const long t = m / p.Wo;
        const int ho = (int)(t % p.Ho);
        const long n = t / p.Ho;
        const int g = co / cout_g;
        float acc = p.bias ? p.bias[co] : 0.f;
        const float* wk = w + (size_t)co * p.kh * p.kw * cin_g;
        for (int r = 0; r < p.kh; ++r) {
            const int th = ho + p.pad_h - r * p.dil_h;
            if (th < 0 || th % p.stride_h) continue;
            const int hi = th / p.stride_h;
            if (hi >= p.H) continue;
            for (int q = 0; q < p.kw; ++q) {
                const int tw = wo + p.pad_w - q * p.dil_w;
                if (tw < 0 || tw % p.stride_w) continue;
                const int wi = tw / p.stride_w;
                if (wi >= p.W) continue;
                const T* ip = in + ((n * p.H + hi) * p.W + wi) * p.ld_in + g * cin_g;
                const float* wp = wk + (r * p.kw + q) * cin_g;
                for (int c = 0; c < cin_g; ++c) acc = fmaf(to_f(ip[c]), wp[c], acc);
            }
        }
        acc = act_f(acc, p.act1, p.alpha1);
        out[m * p.ld_out + co] = from_f<T>(acc);
    }
}

template <typename T>
constexpr int vec_of() {
    return 16 / sizeof(T);
}

inline bool can_vec(int dtype, int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    const int v = dtype == DT_F16 ? 8 : 4;
    if (C % v) return false;
    for (int ld : lds)
        if (ld % v) return false;
    for (const void* p : ptrs)
        if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
    return true;
}

#define DISPATCH_T_V(dtype, vec, CALL)                            \
    do {                                                          \
        if ((dtype) == DT_F16) {                                  \
            if (vec) { CALL(_Float16, 8); } else { CALL(_Float16, 1); } \
        } else {                                                  \
            if (vec) { CALL(float, 4); } else { CALL(float, 1); } \
        }                                                         \
    } while (0)

}  // namespace

int32_t nchw_f32_to_nhwc(const float* in, void* out, int dtype, int N, int C, int H, int W, int Cpad, int ld_out,
                         hipStream_t s) {
    const long total = (long)N * H * W * Cpad;
    if (dtype == DT_F16 && Cpad % 8 == 0 && ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && C >= 16 && N <= 65535) {
        const long HW = (long)H * W;
        hipLaunchKernelGGL(nchw_to_nhwc_f16_tiled_kernel, dim3((unsigned)((HW + 31) / 32), (unsigned)((Cpad + 63) / 64), (unsigned)N),
                           dim3(256), 0, s, in, static_cast<_Float16*>(out), C, HW, Cpad, ld_out);
        return check_launch("nchw_f32_to_nhwc");
    }
    if (dtype == DT_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<_Float16>, dim3(grid_for(total)), dim3(kThreads), 0, s, in,
                           static_cast<_Float16*>(out), N, C, H, W, Cpad, ld_out);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(kThreads), 0, s, in,
                           static_cast<float*>(out), N, C, H, W, Cpad, ld_out);
    return check_launch("nchw_f32_to_nhwc");
}

int32_t nhwc_to_nchw_f32(const void* in, int dtype, float* out, int N, int C, int H, int W, int ld_in, hipStream_t s) {
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    if (dtype == DT_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, grid, dim3(256), 0, s, static_cast<const _Float16*>(in), out,
                           C, HW, ld_in);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(in), out, C, HW,
                           ld_in);
    return check_launch("nhwc_to_nchw_f32");
}

int32_t nhwc_pool(const void* in, void* out, int dtype, int op, int N, int H, int W, int C, int ld_in, int Ho, int Wo,
                  int ld_out, int kh, int kw, int sh, int sw, int ph, int pw, int avg_exclusive, hipStream_t s) {
    const bool vec = can_vec(dtype, C, {ld_in, ld_out}, {in, out});
#define CALL(T, V)                                                                                                  \
    hipLaunchKernelGGL((pool_kernel<T, V>), dim3(grid_for((long)N * Ho * Wo * (C / V))), dim3(kThreads), 0, s,     \
                       static_cast<const T*>(in), static_cast<T*>(out), op, N, H, W, C, ld_in, Ho, Wo, ld_out, kh, \
                       kw, sh, sw, ph, pw, avg_exclusive)
    DISPATCH_T_V(dtype, vec, CALL);
#undef CALL
    return check_launch("nhwc_pool");
}

int32_t nhwc_maxpool_chain3_f16(const void* in, void* o1, void* o2, void* o3, int N, int H, int W, int C, int ld_in, int ld1,
                                int ld2, int ld3, int k, hipStream_t s, int f32) {
    const size_t lds = (size_t)2 * (H + k - 1) * (W + k - 1) * 16;  // two padded planes
    const int V = f32 ? 4 : 8;
    if (C % V || lds > 64 * 1024) return TRTX_ERR_UNSUPPORTED;
    if (f32)
        hipLaunchKernelGGL((maxpool_chain3_kernel<float, 4>), dim3((unsigned)(N * (C / 4))), dim3(256), lds, s, static_cast<const float*>(in),
                           static_cast<float*>(o1), static_cast<float*>(o2), static_cast<float*>(o3), H, W, C, ld_in, ld1, ld2, ld3, k);
    else
        hipLaunchKernelGGL((maxpool_chain3_kernel<_Float16, 8>), dim3((unsigned)(N * (C / 8))), dim3(256), lds, s, static_cast<const _Float16*>(in),
                           static_cast<_Float16*>(o1), static_cast<_Float16*>(o2), static_cast<_Float16*>(o3), H, W, C, ld_in, ld1, ld2,
                           ld3, k);
    return check_launch("nhwc_maxpool_chain3");
}

// depth to space (fp16, 8-channel chunks): out[n][h*bh + r][w*bw + q][c] = in[n][h][w][(r*bw + q)*C + c]
__global__ void depth_to_space_f16_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, long total, int H, int W,
                                          int C, int bh, int bw, int ld_in, int ld_out) {
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cv = C / 8;
    const int c8 = (int)(i % cv);
    long px = i / cv;  // output pixel (n, ho, wo)
    const int Wo = W * bw, Ho = H * bh;
    const int wo = (int)(px % Wo);
    px /= Wo;
    const int ho = (int)(px % Ho);
    const long n = px / Ho;
    const int h = ho / bh, r = ho - h * bh, w = wo / bw, q = wo - w * bw;
    const half8_t v = *reinterpret_cast<const half8_t*>(in + ((n * H + h) * W + w) * ld_in + (r * bw + q) * C + c8 * 8);
    *reinterpret_cast<half8_t*>(out + ((n * Ho + ho) * Wo + wo) * ld_out + c8 * 8) = v;
}

int32_t nhwc_depth_to_space_f16(const void* in, void* out, int N, int H, int W, int C, int bh, int bw, int ld_in, int ld_out,
                                hipStream_t s) {
    if (C % 8 || ld_in % 8 || ld_out % 8) return TRTX_ERR_UNSUPPORTED;
    const long total = (long)N * H * bh * W * bw * (C / 8);
    hipLaunchKernelGGL(depth_to_space_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       static_cast<const _Float16*>(in), static_cast<_Float16*>(out), total, H, W, C, bh, bw, ld_in, ld_out);
    return check_launch("nhwc_depth_to_space_f16");
}

int32_t nhwc_resize_nearest(const void* in, void* out, int dtype, int N, int H, int W, int C, int ld_in, int Ho,
                            int Wo, int ld_out, hipStream_t s) {
    const bool vec = can_vec(dtype, C, {ld_in, ld_out}, {in, out});
#define CALL(T, V)                                                                                                   \
    hipLaunchKernelGGL((resize_nearest_kernel<T, V>), dim3(grid_for((long)N * Ho * Wo * (C / V))), dim3(kThreads), \
                       0, s, static_cast<const T*>(in), static_cast<T*>(out), N, H, W, C, ld_in, Ho, Wo, ld_out)
    DISPATCH_T_V(dtype, vec, CALL);
#undef CALL
    return check_launch("nhwc_resize_nearest");
}

int32_t nhwc_elementwise(const void* a, const void* b, void* out, int dtype, int op, long pixels, int C, int ld_a,
                         int ld_b, int ld_out, hipStream_t s) {
    const bool vec = can_vec(dtype, C, {ld_a, ld_b, ld_out}, {a, b, out});
#define CALL(T, V)                                                                                               \
    hipLaunchKernelGGL((elementwise_kernel<T, V>), dim3(grid_for(pixels * (C / V))), dim3(kThreads), 0, s,      \
                       static_cast<const T*>(a), static_cast<const T*>(b), static_cast<T*>(out), op, pixels, C, \
                       ld_a, ld_b, ld_out)
    DISPATCH_T_V(dtype, vec, CALL);
#undef CALL
    return check_launch("nhwc_elementwise");
}

int32_t nhwc_activation(const void* in, void* out, int dtype, int act, float alpha, long pixels, int C, int ld_in,
                        int ld_out, hipStream_t s) {
    const bool vec = can_vec(dtype, C, {ld_in, ld_out}, {in, out});
#define CALL(T, V)                                                                                          \
    hipLaunchKernelGGL((activation_kernel<T, V>), dim3(grid_for(pixels * (C / V))), dim3(kThreads), 0, s,  \
                       static_cast<const T*>(in), static_cast<T*>(out), act, alpha, pixels, C, ld_in, ld_out)
    DISPATCH_T_V(dtype, vec, CALL);
#undef CALL
    return check_launch("nhwc_activation");
}

int32_t nhwc_scale(const void* in, void* out, int dtype, const float* scale, const float* shift, long pixels, int C,
                   int ld_in, int ld_out, hipStream_t s) {
    const bool vec = can_vec(dtype, C, {ld_in, ld_out}, {in, out});
#define CALL(T, V)                                                                                     \
    hipLaunchKernelGGL((scale_kernel<T, V>), dim3(grid_for(pixels * (C / V))), dim3(kThreads), 0, s,  \
                       static_cast<const T*>(in), static_cast<T*>(out), scale, shift, pixels, C, ld_in, ld_out)
    DISPATCH_T_V(dtype, vec, CALL);
#undef CALL
    return check_launch("nhwc_scale");
}

int32_t nhwc_copy(const void* in, void* out, int dtype, long pixels, int C, int ld_in, int ld_out, hipStream_t s) {
    const bool vec = can_vec(dtype, C, {ld_in, ld_out}, {in, out});
#define CALL(T, V)                                                                                    \
    hipLaunchKernelGGL((copy_kernel<T, V>), dim3(grid_for(pixels * (C / V))), dim3(kThreads), 0, s,  \
                       static_cast<const T*>(in), static_cast<T*>(out), pixels, C, ld_in, ld_out)
    DISPATCH_T_V(dtype, vec, CALL);
#undef CALL
    return check_launch("nhwc_copy");
}

int32_t nhwc_reduce_hw_avg(const void* in, void* out, int dtype, int N, int HW, int C, int ld_in, int ld_out,
                           hipStream_t s) {
    dim3 grid((C + 63) / 64, N);
    if (dtype == DT_F16 && C % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(reduce_hw_avg_f16x8_kernel, dim3((C / 8 + 63) / 64, N), dim3(256), 0, s, static_cast<const _Float16*>(in),
                           static_cast<_Float16*>(out), HW, C, ld_in, ld_out);
    else if (dtype == DT_F16)
        hipLaunchKernelGGL(reduce_hw_avg_kernel<_Float16>, grid, dim3(256), 0, s, static_cast<const _Float16*>(in),
                           static_cast<_Float16*>(out), HW, C, ld_in, ld_out);
    else
        hipLaunchKernelGGL(reduce_hw_avg_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(in),
                           static_cast<float*>(out), HW, C, ld_in, ld_out);
    return check_launch("nhwc_reduce_hw_avg");
}

int32_t conv_direct(const ConvArgs& a, int dtype, hipStream_t s) {
    const long total = (long)a.M * a.Cout;
    if (dtype == DT_F16)
        hipLaunchKernelGGL(conv_direct_kernel<_Float16>, dim3(grid_for(total)), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL(conv_direct_kernel<float>, dim3(grid_for(total)), dim3(kThreads), 0, s, a);
    return check_launch("conv_direct");
}

int32_t deconv_direct(const ConvArgs& a, int dtype, hipStream_t s) {
    const long total = (long)a.M * a.Cout;
    if (dtype == DT_F16)
        hipLaunchKernelGGL(deconv_direct_kernel<_Float16>, dim3(grid_for(total)), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL(deconv_direct_kernel<float>, dim3(grid_for(total)), dim3(kThreads), 0, s, a);
    return check_launch("deconv_direct");
}

}  // namespace trtx
