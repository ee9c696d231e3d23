// YOLOv8 anchor-free decode for gfx950 (MI355X) — deterministic two-pass compaction.
//
// Replaces YoloLayerPlugin::forwardGpu + CalDetection of the reference
// (yolov8/plugin/yololayer.cu:178-220, 282-316).  Same arithmetic per cell:
//   p_c = 1/(1+expf(-logit_c)); argmax with strict '>' from (0.0, class 0); drop if p < 0.1;
//   bbox = [(col+.5-l)*s, (row+.5-t)*s, (col+.5+r)*s, (row+.5+b)*s]; conf = p; class_id = argmax.
// Differences (both documented in DESIGN.md):
//   * slots are handed out in canonical (level, cell) order by a prefix scan instead of atomicAdd,
//     so the output is bit-reproducible (the reference's slot order is a race);
//   * out[b][0] is clamped to max_out (the reference lets the counter run past the buffer).
//
// HBM-bound: reads (4+classes) x cells x 4 B per image once (pass 1, 16 B per lane, coalesced along
// the cell axis), then 8 B/cell of scratch + the 4 box channels of the surviving cells (pass 2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"

namespace {

constexpr int kMaxLevels = 8;
constexpr int kChunk = 512;  // cells per workgroup in both passes

// which optional branches of the YOLOv8 Detection record are decoded (yololayer.cu:222-279)
struct YoloBranches {
    int seg, pose, obb, nk;
    float kpt_conf;
};

struct LevelTable {
    const float* in[kMaxLevels];  // device pointers, [batch][4+classes][cells]
    int cell_off[kMaxLevels + 1];  // cumulative cell offsets
    int grid_w[kMaxLevels];
    int stride[kMaxLevels];
    int n_levels;
};

__device__ __forceinline__ float logist(float x) {
    return 1.0f / (1.0f + expf(-x));
}

__device__ __forceinline__ int find_level(const LevelTable& t, int g) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < t.n_levels && g >= t.cell_off[i])
            l = i;
    return l;
}

// Pass 1: per cell best class/prob.  VEC cells per thread (VEC = 4 -> 16-byte loads).
template <int VEC>
__global__ __launch_bounds__(kChunk / VEC) void yolo_score_kernel(LevelTable t, int classes, int info_len, int total_cells,
                                                                  float* __restrict__ score,
                                                                  int* __restrict__ cls_out,
                                                                  int* __restrict__ chunk_cnt, int n_chunks) {
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int g0 = chunk * kChunk + threadIdx.x * VEC;
    int nflag = 0;
    if (g0 < total_cells) {
        const int l = find_level(t, g0);
        const int cells = t.cell_off[l + 1] - t.cell_off[l];
        const int e0 = g0 - t.cell_off[l];
        const float* cur = t.in[l] + (size_t)b * cells * info_len + e0;
        float best[VEC];
        int bcls[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            best[v] = 0.0f;
            bcls[v] = 0;
        }
#pragma unroll 4
        for (int c = 0; c < classes; ++c) {
            float x[VEC];
            const float* p = cur + (size_t)(4 + c) * cells;
            if constexpr (VEC == 4) {
                const float4 q = *reinterpret_cast<const float4*>(p);
                x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) x[v] = p[v];
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float pr = logist(x[v]);
                if (pr > best[v]) {
                    best[v] = pr;
                    bcls[v] = c;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int g = g0 + v;
            if (g < total_cells) {
                const bool keep = !((double)best[v] < 0.1);  // same comparison as yololayer.cu:203
                score[(size_t)b * total_cells + g] = keep ? best[v] : -1.0f;
                cls_out[(size_t)b * total_cells + g] = bcls[v];
                nflag += keep ? 1 : 0;
            }
        }
    }
    // workgroup reduction of nflag
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int w = nflag;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&s_cnt, w);
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[b * n_chunks + chunk] = s_cnt;
}

// Pass 2: ordered compaction.  One thread per cell, kChunk threads per workgroup.
// `boxes` != nullptr: ltrb distances were already produced by the fused DFL pass ([batch][cells][4]).
__global__ __launch_bounds__(kChunk) void yolo_emit_kernel(LevelTable t, int classes, int total_cells,
                                                           const float* __restrict__ score,
                                                           const int* __restrict__ cls_in,
                                                           const int* __restrict__ chunk_cnt, int n_chunks,
                                                           int max_out, int out_elem, float* __restrict__ output,
                                                           const float4* __restrict__ boxes, YoloBranches br) {
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int kWaves = kChunk / 64;
    __shared__ int s_wave[kWaves];
    __shared__ int s_base;

    // slots used by earlier chunks of this image
    if (wave == 0) {
        int acc = 0;
        for (int j = lane; j < chunk; j += 64) acc += chunk_cnt[b * n_chunks + j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) s_base = acc;
    }
    const int g = chunk * kChunk + threadIdx.x;
    float sc = -1.0f;
    if (g < total_cells) sc = score[(size_t)b * total_cells + g];
    const bool keep = sc >= 0.0f;
    const unsigned long long m = __ballot(keep);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int before = s_base;
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv)
        if (wv < wave) before += s_wave[wv];
    const int slot = before + in_wave;
    float* out = output + (size_t)b * out_elem;
    if (keep && slot < max_out) {
        const int l = find_level(t, g);
        const int cells = t.cell_off[l + 1] - t.cell_off[l];
        const int e = g - t.cell_off[l];
        const int gw = t.grid_w[l];
        const float stride = (float)t.stride[l];
        float4 d;
        const int info_len = 4 + classes + (br.seg ? 32 : 0) + (br.pose ? br.nk * 3 : 0) + (br.obb ? 1 : 0);
        const float* cur = boxes ? nullptr : t.in[l] + (size_t)b * cells * info_len + e;
        if (boxes) {
            d = boxes[(size_t)b * total_cells + g];
        } else {
            d = make_float4(cur[0], cur[(size_t)cells], cur[(size_t)2 * cells], cur[(size_t)3 * cells]);
        }
        const int row = e / gw, col = e - row * gw;
        float* det = out + 1 + (size_t)slot * trtx::kYoloDetFloats;
        det[0] = (col + 0.5f - d.x) * stride;
        det[1] = (row + 0.5f - d.y) * stride;
        det[2] = (col + 0.5f + d.z) * stride;
        det[3] = (row + 0.5f + d.w) * stride;
        det[4] = sc;
        det[5] = (float)cls_in[(size_t)b * total_cells + g];
        // ---- the seg / pose / obb branches of CalDetection (yololayer.cu:222-279); plugin-ABI form only
        if (cur && br.seg) {
            const float* m = cur + (size_t)(4 + classes + (br.pose ? br.nk * 3 : 0) + (br.obb ? 1 : 0)) * cells;
            for (int k = 0; k < 32; ++k) det[6 + k] = m[(size_t)k * cells];
        }
        if (cur && br.pose) {
            const int istride = t.stride[l];
            const float* kp = cur + (size_t)(4 + classes + (br.seg ? 32 : 0) + (br.obb ? 1 : 0)) * cells;
            for (int k = 0; k < br.nk; ++k) {
                const float kconf = 1.0f / (1.0f + expf(-kp[(size_t)(k * 3 + 2) * cells]));
                // "(in * 2.0 + col) * stride": double arithmetic, rounded once to float (yololayer.cu:236-237)
                const float kx = (float)((kp[(size_t)(k * 3) * cells] * 2.0 + col) * istride);
                const float ky = (float)((kp[(size_t)(k * 3 + 1) * cells] * 2.0 + row) * istride);
                const bool inside = kx >= det[0] && kx <= det[2] && ky >= det[1] && ky <= det[3];
                float* o = det + 38 + k * 3;
                if (kconf < br.kpt_conf || !inside) {
                    o[0] = -1;
                    o[1] = -1;
                    o[2] = -1;
                } else {
                    o[0] = kx;
                    o[1] = ky;
                    o[2] = kconf;
                }
            }
        }
        if (cur && br.obb) {
            const int istride = t.stride[l];
            const double pi = 3.14159265358979323846;  // M_PI
            const float ain = cur[(size_t)(4 + classes + (br.seg ? 32 : 0) + (br.pose ? br.nk * 3 : 0)) * cells];
            const double angle = ((1.0f / (1.0f + expf(-ain))) - 0.25f) * pi;  // float difference times double pi (:259)
            const double cos1 = cos(angle), sin1 = sin(angle);
            const float xf = (d.z - d.x) / 2, yf = (d.w - d.y) / 2;
            const double x = xf * cos1 - yf * sin1;
            const double y = xf * sin1 + yf * cos1;
            det[0] = (float)((col + 0.5f + x) * istride);
            det[1] = (float)((row + 0.5f + y) * istride);
            det[2] = (d.x + d.z) * istride;
            det[3] = (d.y + d.w) * istride;
            det[trtx::kYoloDetFloats - 1] = (float)angle;
        }
    }
    if (chunk == n_chunks - 1 && threadIdx.x == kChunk - 1) {
        int total = before + in_wave + (keep ? 1 : 0);  // last thread of the last chunk sees the full count
        out[0] = (float)(total < max_out ? total : max_out);
    }
}


// ---------------------------------------------------------------------------------------------------------
// Fused detect-head tail: reads the NHWC fp16 output of the head convolutions directly
// (channels [0,64) = 4 sides x 16 DFL bins, [64, 64+classes) = class logits) and performs, per cell,
//   DFL: softmax over the 16 bins of each side, expectation with the 1x1 "dfl.conv" weights
//        (yolov8/src/block.cpp:239-257 — shuffle/softmax/conv/shuffle collapsed into registers, fp32), and
//   the CalDetection class scan (sigmoid, strict-'>' argmax, 0.1 threshold; yololayer.cu:195-204).
// It replaces ~10 layout/shuffle/slice/softmax/concat launches per level of the un-fused graph.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

struct HeadTable {
    const void* in[kMaxLevels];
    int ld[kMaxLevels];
    int cell_off[kMaxLevels + 1];
    int n_levels;
};

// eight consecutive channels of a cell as floats: one 16-byte load of an fp16 engine's tensor, two of an fp32 engine's
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&x)[8]) {
    if constexpr (sizeof(T) == 2) {
        const half8_t v = *reinterpret_cast<const half8_t*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (float)v[i];
    } else {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
        x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
}

// T = element type of the NHWC head tensors: _Float16 (kFP16 engines) or float (fp32 engines, round 5: the build that meets BASELINE's tolerance
// had run the un-fused graph, ~30 layout / shuffle / softmax launches)
template <typename T>
__global__ __launch_bounds__(256) void yolo_head_score_kernel(HeadTable t, int classes, int total_cells,
                                                              const float* __restrict__ dfl_w,
                                                              float* __restrict__ score, int* __restrict__ cls_out,
                                                              float4* __restrict__ boxes,
                                                              int* __restrict__ chunk_cnt, int n_chunks) {
    __shared__ int s_list[256];
    __shared__ int s_n, s_keep;
    const int b = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) {
        s_n = 0;
        s_keep = 0;
    }
    __syncthreads();
    auto cell_ptr = [&](int gg) {
        int l = 0;
#pragma unroll
        for (int i = 1; i < kMaxLevels; ++i)
            if (i < t.n_levels && gg >= t.cell_off[i]) l = i;
        const int cells = t.cell_off[l + 1] - t.cell_off[l];
        return static_cast<const T*>(t.in[l]) + ((size_t)b * cells + (gg - t.cell_off[l])) * t.ld[l];
    };
    // ---- phase 1, every cell: the cell can only survive if some sigmoid(logit) >= 0.1, i.e. some logit >= -2.1972.  The
    // maximum of the raw fp16 logits (exact, packed max, no exp) settles that; possible survivors are compacted into a
    // list so that the expensive exact pass below runs on dense lanes instead of a few lanes of every wave.
    // The class logits are read by the WAVE, not by the cell's own thread: consecutive lanes take consecutive 16-byte pieces of
    // the 64 cells of the wave (a cell's logits are classes/8 pieces in a row), so one load instruction touches ~15 cache lines
    // instead of 64 (one per lane, 288 bytes apart) - this pass is what the kernel's time is.  A piece whose packed maximum
    // passes the test raises its cell's flag in LDS; NaN pieces do not, exactly like NaN fails 'pr > best' in the scan.
    s_list[threadIdx.x] = 0;   // phase 1 borrows the list as the per-cell flags (same-value races only)
    __syncthreads();
    {
        const int wave0 = threadIdx.x & ~63, lane = threadIdx.x & 63;
        const int pieces = classes >> 3;                 // classes % 8 == 0 on this path
        const int g0 = blockIdx.x * 256 + wave0;
        int cells_here = total_cells - g0;
        cells_here = cells_here > 64 ? 64 : cells_here;
        const int n = cells_here * pieces;
        for (int j = lane; j < n; j += 64) {
            const int c = j / pieces, q = j - c * pieces;
            float v[8];
            load8(cell_ptr(g0 + c) + 64 + q * 8, v);
            float m = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
            if (m > -2.3f) s_list[wave0 + c] = 1;
        }
    }
    __syncthreads();
    bool maybe = false;
    if (g < total_cells) {
        maybe = s_list[threadIdx.x] != 0;
        if (!maybe) {
            const size_t o = (size_t)b * total_cells + g;
            score[o] = -1.0f;
            cls_out[o] = 0;
        }
    }
    __syncthreads();   // the flags are read; the list proper is written next
    {
        const unsigned long long m = __ballot(maybe);
        int base = 0;
        if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(&s_n, __popcll(m));
        base = __shfl(base, 0);
        if (maybe) s_list[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = threadIdx.x;
    }
    __syncthreads();
    // ---- phase 2, possible survivors only: the reference arithmetic (DFL softmax . w per side, sigmoid of every class, strict
    // '>' argmax), so kept candidates are bit-identical to evaluating every cell
    int kept = 0;
    for (int k = threadIdx.x; k < s_n; k += 256) {
        const int gg = blockIdx.x * 256 + s_list[k];
        const T* cell = cell_ptr(gg);
        float w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = dfl_w[i];
        float side[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float lo[8], hi[8], x[16];
            load8(cell + s * 16, lo);
            load8(cell + s * 16 + 8, hi);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[i] = lo[i];
                x[8 + i] = hi[i];
            }
            float mx = x[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, x[i]);
            float sum = 0.f, acc = 0.f;
            if constexpr (sizeof(T) == 4) {
                // fp32 engines: the arithmetic of the un-fused chain to the bit - softmax_kernel (linear_ops.hip: p_i = expf(x_i - max) * (1 / sum)) followed by the
                // DFL 1x1 convolution (conv_direct_kernel: an fmaf chain over the 16 bins from 0) - so that a user YoloLayer plugin behind the un-fused graph and
                // the built-in fused tail return the same boxes (tests/test_ref_pinning.py::test_reference_yololayer_plugin_runs_inside_a_full_engine)
                float ex[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    ex[i] = expf(x[i] - mx);
                    sum += ex[i];
                }
                const float inv = 1.0f / sum;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc = fmaf(ex[i] * inv, w[i], acc);
                side[s] = acc;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float ex = expf(x[i] - mx);
                    sum += ex;
                    acc = fmaf(ex, w[i], acc);
                }
                side[s] = acc / sum;
            }
        }
        float best = 0.0f;
        int bcls = 0;
        const T* cl = cell + 64;
        for (int c0 = 0; c0 < classes; c0 += 8) {  // classes % 8 == 0 on this path
            float v[8];
            load8(cl + c0, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float pr = logist(v[i]);
                if (pr > best) {
                    best = pr;
                    bcls = c0 + i;
                }
            }
        }
        const bool keep = !((double)best < 0.1);
        const size_t o = (size_t)b * total_cells + gg;
        score[o] = keep ? best : -1.0f;
        cls_out[o] = bcls;
        boxes[o] = make_float4(side[0], side[1], side[2], side[3]);
        kept += keep ? 1 : 0;
    }
    // per-kChunk candidate counts (two 256-thread workgroups feed one chunk counter)
    if (kept) atomicAdd(&s_keep, kept);
    __syncthreads();
    if (threadIdx.x == 0 && s_keep) atomicAdd(&chunk_cnt[b * n_chunks + (blockIdx.x * 256) / kChunk], s_keep);
}

}  // namespace

extern "C" size_t trtx_yolo_decode_workspace(int batch, int net_h, int net_w, const int* strides, int n_levels) {
    size_t cells = 0;
    for (int i = 0; i < n_levels; ++i) cells += (size_t)(net_h / strides[i]) * (net_w / strides[i]);
    const size_t n_chunks = (cells + kChunk - 1) / kChunk;
    return trtx::align_up((size_t)batch * cells * sizeof(float), 256) +
           trtx::align_up((size_t)batch * cells * sizeof(int), 256) +
           trtx::align_up((size_t)batch * n_chunks * sizeof(int), 256);
}

extern "C" int32_t trtx_yolo_decode_ex(const float* const* inputs, int n_levels, int batch, int classes, int net_h, int net_w,
                                       const int* strides, int max_out, int n_kpt, float kpt_conf, int is_seg, int is_pose, int is_obb,
                                       float* output, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (n_levels < 1 || n_levels > kMaxLevels || batch < 1 || classes < 1 || max_out < 1 || !inputs || !output ||
        !workspace || n_kpt < 0 || n_kpt > 17)
        return TRTX_ERR_INVALID;
    if (workspace_bytes < trtx_yolo_decode_workspace(batch, net_h, net_w, strides, n_levels)) return TRTX_ERR_WORKSPACE;
    const YoloBranches br{is_seg ? 1 : 0, is_pose ? 1 : 0, is_obb ? 1 : 0, n_kpt, kpt_conf};
    const int info_len = 4 + classes + (br.seg ? 32 : 0) + (br.pose ? n_kpt * 3 : 0) + (br.obb ? 1 : 0);
    LevelTable t{};
    t.n_levels = n_levels;
    bool vec4 = true;
    int off = 0;
    for (int i = 0; i < n_levels; ++i) {
        const int gh = net_h / strides[i], gw = net_w / strides[i];
        t.in[i] = inputs[i];
        t.cell_off[i] = off;
        t.grid_w[i] = gw;
        t.stride[i] = strides[i];
        off += gh * gw;
        if ((gh * gw) % 4 != 0 || (reinterpret_cast<uintptr_t>(inputs[i]) & 15) != 0) vec4 = false;
    }
    for (int i = n_levels; i <= kMaxLevels; ++i) t.cell_off[i] = off;
    const int total_cells = off;
    const int n_chunks = (total_cells + kChunk - 1) / kChunk;
    char* ws = static_cast<char*>(workspace);
    float* score = reinterpret_cast<float*>(ws);
    ws += trtx::align_up((size_t)batch * total_cells * sizeof(float), 256);
    int* cls = reinterpret_cast<int*>(ws);
    ws += trtx::align_up((size_t)batch * total_cells * sizeof(int), 256);
    int* chunk_cnt = reinterpret_cast<int*>(ws);
    const int out_elem = 1 + max_out * trtx::kYoloDetFloats;
    dim3 grid(n_chunks, batch);
    if (vec4)
        hipLaunchKernelGGL(yolo_score_kernel<4>, grid, dim3(kChunk / 4), 0, stream, t, classes, info_len, total_cells, score,
                           cls, chunk_cnt, n_chunks);
    else
        hipLaunchKernelGGL(yolo_score_kernel<1>, grid, dim3(kChunk), 0, stream, t, classes, info_len, total_cells, score, cls,
                           chunk_cnt, n_chunks);
    hipLaunchKernelGGL(yolo_emit_kernel, grid, dim3(kChunk), 0, stream, t, classes, total_cells, score, cls,
                       chunk_cnt, n_chunks, max_out, out_elem, output, (const float4*)nullptr, br);
    return trtx::check_launch("trtx_yolo_decode");
}

extern "C" int32_t trtx_yolo_decode(const float* const* inputs, int n_levels, int batch, int classes, int net_h,
                                    int net_w, const int* strides, int max_out, float* output, void* workspace,
                                    size_t workspace_bytes, hipStream_t stream) {
    return trtx_yolo_decode_ex(inputs, n_levels, batch, classes, net_h, net_w, strides, max_out, 0, 0.f, 0, 0, 0, output, workspace,
                               workspace_bytes, stream);
}


extern "C" size_t trtx_yolo_head_decode_workspace(int batch, int net_h, int net_w, const int* strides, int n_levels) {
    size_t cells = 0;
    for (int i = 0; i < n_levels; ++i) cells += (size_t)(net_h / strides[i]) * (net_w / strides[i]);
    const size_t n_chunks = (cells + kChunk - 1) / kChunk;
    return trtx::align_up((size_t)batch * cells * sizeof(float), 256) +
           trtx::align_up((size_t)batch * cells * sizeof(int), 256) +
           trtx::align_up((size_t)batch * cells * sizeof(float4), 256) +
           trtx::align_up((size_t)batch * n_chunks * sizeof(int), 256);
}

static int32_t head_decode(const void* const* heads, const int* ld, int elem_bytes, int n_levels, int batch, int classes, int net_h, int net_w, const int* strides,
                           const float* dfl_weights, int max_out, float* output, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (n_levels < 1 || n_levels > kMaxLevels || batch < 1 || classes < 8 || classes % 8 || max_out < 1 || !heads ||
        !ld || !dfl_weights || !output || !workspace)
        return TRTX_ERR_INVALID;
    if (workspace_bytes < trtx_yolo_head_decode_workspace(batch, net_h, net_w, strides, n_levels)) return TRTX_ERR_WORKSPACE;
    HeadTable h{};
    LevelTable t{};
    h.n_levels = t.n_levels = n_levels;
    int off = 0;
    for (int i = 0; i < n_levels; ++i) {
        const int gh = net_h / strides[i], gw = net_w / strides[i];
        if (ld[i] % (16 / elem_bytes) || (reinterpret_cast<uintptr_t>(heads[i]) & 15)) return TRTX_ERR_UNSUPPORTED;
        h.in[i] = heads[i];
        h.ld[i] = ld[i];
        h.cell_off[i] = t.cell_off[i] = off;
        t.grid_w[i] = gw;
        t.stride[i] = strides[i];
        off += gh * gw;
    }
    for (int i = n_levels; i <= kMaxLevels; ++i) h.cell_off[i] = t.cell_off[i] = off;
    const int total_cells = off;
    const int n_chunks = (total_cells + kChunk - 1) / kChunk;
    char* ws = static_cast<char*>(workspace);
    float* score = reinterpret_cast<float*>(ws);
    ws += trtx::align_up((size_t)batch * total_cells * sizeof(float), 256);
    int* cls = reinterpret_cast<int*>(ws);
    ws += trtx::align_up((size_t)batch * total_cells * sizeof(int), 256);
    float4* boxes = reinterpret_cast<float4*>(ws);
    ws += trtx::align_up((size_t)batch * total_cells * sizeof(float4), 256);
    int* chunk_cnt = reinterpret_cast<int*>(ws);
    if (hipMemsetAsync(chunk_cnt, 0, (size_t)batch * n_chunks * sizeof(int), stream) != hipSuccess) return TRTX_ERR_HIP;
    const int out_elem = 1 + max_out * trtx::kYoloDetFloats;
    if (elem_bytes == 2)
        hipLaunchKernelGGL(yolo_head_score_kernel<_Float16>, dim3((total_cells + 255) / 256, batch), dim3(256), 0, stream, h, classes, total_cells, dfl_weights,
                           score, cls, boxes, chunk_cnt, n_chunks);
    else
        hipLaunchKernelGGL(yolo_head_score_kernel<float>, dim3((total_cells + 255) / 256, batch), dim3(256), 0, stream, h, classes, total_cells, dfl_weights,
                           score, cls, boxes, chunk_cnt, n_chunks);
    hipLaunchKernelGGL(yolo_emit_kernel, dim3(n_chunks, batch), dim3(kChunk), 0, stream, t, classes, total_cells, score,
                       cls, chunk_cnt, n_chunks, max_out, out_elem, output, (const float4*)boxes, YoloBranches{0, 0, 0, 0, 0.f});
    return trtx::check_launch("trtx_yolo_head_decode_nhwc");
}

extern "C" int32_t trtx_yolo_head_decode_nhwc(const void* const* heads, const int* ld, int n_levels, int batch,
                                              int classes, int net_h, int net_w, const int* strides,
                                              const float* dfl_weights, int max_out, float* output, void* workspace,
                                              size_t workspace_bytes, hipStream_t stream) {
    return head_decode(heads, ld, 2, n_levels, batch, classes, net_h, net_w, strides, dfl_weights, max_out, output, workspace, workspace_bytes, stream);
}

extern "C" int32_t trtx_yolo_head_decode_nhwc_f32(const void* const* heads, const int* ld, int n_levels, int batch, int classes, int net_h, int net_w,
                                                  const int* strides, const float* dfl_weights, int max_out, float* output, void* workspace,
                                                  size_t workspace_bytes, hipStream_t stream) {
    return head_decode(heads, ld, 4, n_levels, batch, classes, net_h, net_w, strides, dfl_weights, max_out, output, workspace, workspace_bytes, stream);
}
