// Anchor-based YoloLayer decode (YOLOv5 / v7 / v3-v4 family) for gfx950 (MI355X) — deterministic two-pass compaction.
//
// Replaces YoloLayerPlugin::forwardGpu + CalDetection of the reference (yolov5/plugin/yololayer.cu:161-227).  Per grid cell
// and per anchor k of a level (input [batch][3 * (5 + classes (+32))][cells], channel-major):
//   box_prob = sigmoid(obj); dropped if box_prob < kIgnoreThresh (0.1f, yolov5/src/config.h:38);
//   class scan: p = sigmoid(logit_c), strict '>' from (0.0, class 0);
//   bbox = [(col - 0.5 + 2 sigma(x)) * netW / gridW, (row - 0.5 + 2 sigma(y)) * netH / gridH,
//           (2 sigma(w))^2 * anchor_w, (2 sigma(h))^2 * anchor_h]           (centre format)
//   conf = box_prob * max class prob; class_id; 32 mask coefficients copied when is_segmentation.
// Records are Detection structs of 38 floats (yolov5/src/types.h:11-16).
// As in yolo_decode.hip the reference's atomicAdd slot race is replaced by a canonical order — (level, cell, anchor)
// ascending, handed out by a prefix scan — and out[b][0] is clamped to max_out.
//
// HBM-bound: pass 1 reads the objectness plane of every anchor (12 B per cell) and the class planes of the few cells that pass;
// pass 2 touches only survivors.  One thread per cell in both passes (coalesced along the cell axis), three anchors each.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common.h"

namespace {

constexpr int kMaxLevels = 8;
constexpr int kAnchors = 3;   // kNumAnchor, yolov5/src/config.h:35
constexpr int kChunk = 512;   // cells per workgroup
constexpr int kDet5 = 38;     // sizeof(Detection) / 4: bbox[4], conf, class_id, mask[32]

struct Level5Table {
    const float* in[kMaxLevels];
    int cell_off[kMaxLevels + 1];
    int grid_w[kMaxLevels], grid_h[kMaxLevels];
    float anchors[kMaxLevels][kAnchors * 2];
    int n_levels;
};

__device__ __forceinline__ float logist(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ int find_level(const Level5Table& t, int g) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < t.n_levels && g >= t.cell_off[i]) l = i;
    return l;
}

// Pass 1: conf / class of every (cell, anchor); -1 marks a dropped candidate.  score / cls: [batch][cells][3].
__global__ __launch_bounds__(kChunk) void yolo5_score_kernel(Level5Table t, int classes, int info_len, int total_cells,
                                                             float* __restrict__ score,
                                                             int* __restrict__ cls_out, int* __restrict__ chunk_cnt, int n_chunks) {
    const int b = blockIdx.y;
    const int g = blockIdx.x * kChunk + threadIdx.x;
    int nkeep = 0;
    if (g < total_cells) {
        const int l = find_level(t, g);
        const int cells = t.cell_off[l + 1] - t.cell_off[l];
        const int e = g - t.cell_off[l];
        const float* cur = t.in[l] + (size_t)b * info_len * cells * kAnchors + e;
#pragma unroll
        for (int k = 0; k < kAnchors; ++k) {
            const float* a = cur + (size_t)k * info_len * cells;
            const float box_prob = logist(a[(size_t)4 * cells]);
            float conf = -1.0f;
            int best_c = 0;
            if (!(box_prob < 0.1f)) {  // "if (box_prob < kIgnoreThresh) continue;": NaN is kept, as there
                float best = 0.0f;
                for (int c = 0; c < classes; ++c) {
                    const float p = logist(a[(size_t)(5 + c) * cells]);
                    if (p > best) {
                        best = p;
                        best_c = c;
                    }
                }
                conf = box_prob * best;
                // conf >= 0 marks "kept" below; a NaN product (NaN logits) must stay a kept record as in the reference
                if (!(conf >= 0.0f)) conf = __builtin_nanf("");
                ++nkeep;
            }
            const size_t o = ((size_t)b * total_cells + g) * kAnchors + k;
            score[o] = conf;
            cls_out[o] = best_c;
        }
    }
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int w = nkeep;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&s_cnt, w);
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[b * n_chunks + blockIdx.x] = s_cnt;
}

// Pass 2: ordered compaction, one thread per cell (0..3 records each).
__global__ __launch_bounds__(kChunk) void yolo5_emit_kernel(Level5Table t, int classes, int info_len, int total_cells, int net_w, int net_h,
                                                            int is_seg, const float* __restrict__ score, const int* __restrict__ cls_in,
                                                            const int* __restrict__ chunk_cnt, int n_chunks, int max_out, int out_elem,
                                                            float* __restrict__ output) {
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int kWaves = kChunk / 64;
    __shared__ int s_wave[kWaves];
    __shared__ int s_base;
    if (wave == 0) {
        int acc = 0;
        for (int j = lane; j < chunk; j += 64) acc += chunk_cnt[b * n_chunks + j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) s_base = acc;
    }
    const int g = chunk * kChunk + threadIdx.x;
    float sc[kAnchors] = {-1.0f, -1.0f, -1.0f};
    int mine = 0;
    if (g < total_cells) {
#pragma unroll
        for (int k = 0; k < kAnchors; ++k) {
            sc[k] = score[((size_t)b * total_cells + g) * kAnchors + k];
            mine += (sc[k] >= 0.0f || sc[k] != sc[k]) ? 1 : 0;  // kept: conf >= 0 or NaN
        }
    }
    // exclusive prefix of `mine` inside the wave
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int slot = s_base + incl - mine;
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv)
        if (wv < wave) slot += s_wave[wv];
    float* out = output + (size_t)b * out_elem;
    if (mine) {
        const int l = find_level(t, g);
        const int cells = t.cell_off[l + 1] - t.cell_off[l];
        const int e = g - t.cell_off[l];
        const int gw = t.grid_w[l], gh = t.grid_h[l];
        const int row = e / gw, col = e - row * gw;
        const float* cur = t.in[l] + (size_t)b * info_len * cells * kAnchors + e;
#pragma unroll
        for (int k = 0; k < kAnchors; ++k) {
            if (!(sc[k] >= 0.0f || sc[k] != sc[k])) continue;
            if (slot < max_out) {
                const float* a = cur + (size_t)k * info_len * cells;
                float* det = out + 1 + (size_t)slot * kDet5;
                // yololayer.cu:199-206, operation for operation
                det[0] = (col - 0.5f + 2.0f * logist(a[0])) * net_w / gw;
                det[1] = (row - 0.5f + 2.0f * logist(a[(size_t)cells])) * net_h / gh;
                float bw = 2.0f * logist(a[(size_t)2 * cells]);
                bw = bw * bw * t.anchors[l][2 * k];
                float bh = 2.0f * logist(a[(size_t)3 * cells]);
                bh = bh * bh * t.anchors[l][2 * k + 1];
                det[2] = bw;
                det[3] = bh;
                det[4] = sc[k];
                det[5] = (float)cls_in[((size_t)b * total_cells + g) * kAnchors + k];
                if (is_seg)
                    for (int i = 0; i < 32; ++i) det[6 + i] = a[(size_t)(5 + classes + i) * cells];
            }
            ++slot;
        }
    }
    if (chunk == n_chunks - 1 && threadIdx.x == kChunk - 1) {
        const int total = s_base + [&] {
            int tsum = 0;
            for (int wv = 0; wv < kWaves; ++wv) tsum += s_wave[wv];
            return tsum;
        }();
        out[0] = (float)(total < max_out ? total : max_out);
    }
}

}  // namespace

extern "C" size_t trtx_yolov5_decode_workspace(int batch, const int* grid_w, const int* grid_h, int n_levels) {
    size_t cells = 0;
    for (int i = 0; i < n_levels; ++i) cells += (size_t)grid_w[i] * grid_h[i];
    const size_t n_chunks = (cells + kChunk - 1) / kChunk;
    return 2 * trtx::align_up((size_t)batch * cells * kAnchors * 4, 256) + trtx::align_up((size_t)batch * n_chunks * sizeof(int), 256);
}

extern "C" int32_t trtx_yolov5_decode(const float* const* inputs, int n_levels, int batch, int classes, int net_h, int net_w,
                                      const int* grid_w, const int* grid_h, const float* anchors, int max_out, int is_segmentation,
                                      float* output, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (n_levels < 1 || n_levels > kMaxLevels || batch < 1 || classes < 1 || max_out < 1 || !inputs || !grid_w || !grid_h || !anchors ||
        !output || !workspace)
        return TRTX_ERR_INVALID;
    if (workspace_bytes < trtx_yolov5_decode_workspace(batch, grid_w, grid_h, n_levels)) return TRTX_ERR_WORKSPACE;
    Level5Table t{};
    t.n_levels = n_levels;
    int off = 0;
    for (int i = 0; i < n_levels; ++i) {
        if (grid_w[i] < 1 || grid_h[i] < 1 || !inputs[i]) return TRTX_ERR_INVALID;
        t.in[i] = inputs[i];
        t.cell_off[i] = off;
        t.grid_w[i] = grid_w[i];
        t.grid_h[i] = grid_h[i];
        for (int k = 0; k < kAnchors * 2; ++k) t.anchors[i][k] = anchors[i * kAnchors * 2 + k];
        off += grid_w[i] * grid_h[i];
    }
    for (int i = n_levels; i <= kMaxLevels; ++i) t.cell_off[i] = off;
    const int total_cells = off;
    const int n_chunks = (total_cells + kChunk - 1) / kChunk;
    const int info_len = 5 + classes + (is_segmentation ? 32 : 0);
    char* ws = static_cast<char*>(workspace);
    const size_t plane = trtx::align_up((size_t)batch * total_cells * kAnchors * 4, 256);
    float* score = reinterpret_cast<float*>(ws);
    int* cls = reinterpret_cast<int*>(ws + plane);
    int* chunk_cnt = reinterpret_cast<int*>(ws + 2 * plane);
    const int out_elem = 1 + max_out * kDet5;
    const dim3 grid(n_chunks, batch);
    hipLaunchKernelGGL(yolo5_score_kernel, grid, dim3(kChunk), 0, stream, t, classes, info_len, total_cells, score, cls, chunk_cnt, n_chunks);
    hipLaunchKernelGGL(yolo5_emit_kernel, grid, dim3(kChunk), 0, stream, t, classes, info_len, total_cells, net_w, net_h, is_segmentation,
                       score, cls, chunk_cnt, n_chunks, max_out, out_elem, output);
    return trtx::check_launch("trtx_yolov5_decode");
}
