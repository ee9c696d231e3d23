// Host side of the implicit-GEMM convolution family (fp16 / int8): launchers, tactic lists, the grouped launch, the wave-split-K kernel.
// The tile function, its epilogues and the two kernel entry points live in igemm_tile.h (shared with conv_igemm_f32.hip).
#include "../options.h"
#include "igemm_tile.h"
#include "patch_tile.h"

namespace trtx {
namespace {


// Grouped launch: 2..4 independent layers of one instantiation (the detect head's second 3x3 of every level: 64 -> 64 and 80 -> 80 at 80 / 40 / 20
// pixels) in one grid; the tile -> problem mapping of conv_igemm_group_f16_kernel (per-problem XCD chunks, the host orders the problems).
struct ConvPatchGroupArgs {
    int n;
    int slot_start[kMaxConvGroup + 1];
    int chunk[kMaxConvGroup], tiles[kMaxConvGroup], tiles_n[kMaxConvGroup], tiles_x[kMaxConvGroup], tiles_y[kMaxConvGroup];
    unsigned in_bytes[kMaxConvGroup], w_bytes[kMaxConvGroup];
    ConvArgs p[kMaxConvGroup];
};
template <int NFRAG, int KC, int MI>
__global__ __launch_bounds__(256) void conv_patch_group_f16_kernel(const ConvPatchGroupArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[patch_lds_bytes<NFRAG, KC, MI>()];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int pid = 0;
#pragma unroll
    for (int k = 1; k < kMaxConvGroup; ++k) pid += (k < g.n && slot >= g.slot_start[k]) ? 1 : 0;
    pid = __builtin_amdgcn_readfirstlane(pid);
    const int local = xcd * g.chunk[pid] + (slot - g.slot_start[pid]);
    if (local >= g.tiles[pid]) return;
    conv_patch_tile<NFRAG, KC, MI>(g.p[pid], g.in_bytes[pid], g.w_bytes[pid],
                                   patchidx::tile_of(local, g.tiles_n[pid], g.tiles_x[pid], g.tiles_y[pid], 4 * MI, 16 * NFRAG), smem);
}

template <int NFRAG, int KC>
void launch_patch_mi(const ConvArgs& a, unsigned in_bytes, unsigned w_bytes, int th, hipStream_t s) {
    const int tiles_n = a.Cout_pad / (16 * NFRAG), tiles_x = (a.W + 15) / 16, tiles_y = (a.H + th - 1) / th;
    const int total = a.N * tiles_y * tiles_x * tiles_n, chunk = (total + 7) / 8;
    if constexpr (KC <= 4) {   // 16-row tiles exist up to four planes (LDS)
        if (th == 16) {
            TRTX_LAUNCH((conv_patch_f16_kernel<NFRAG, KC, 4>), dim3(chunk * 8), dim3(256), 0, s, a, in_bytes, w_bytes, tiles_n, tiles_x, tiles_y, total, chunk);
            return;
        }
    }
    TRTX_LAUNCH((conv_patch_f16_kernel<NFRAG, KC, 2>), dim3(chunk * 8), dim3(256), 0, s, a, in_bytes, w_bytes, tiles_n, tiles_x, tiles_y, total, chunk);
}
template <int NFRAG>
int32_t launch_patch_kc(const ConvArgs& a, unsigned in_bytes, unsigned w_bytes, int th, hipStream_t s) {
    switch (a.CinK / 32) {
        case 1: launch_patch_mi<NFRAG, 1>(a, in_bytes, w_bytes, th, s); break;
        case 2: launch_patch_mi<NFRAG, 2>(a, in_bytes, w_bytes, th, s); break;
        case 3: launch_patch_mi<NFRAG, 3>(a, in_bytes, w_bytes, th, s); break;
        case 4: launch_patch_mi<NFRAG, 4>(a, in_bytes, w_bytes, th, s); break;
        case 8: launch_patch_mi<NFRAG, 8>(a, in_bytes, w_bytes, 8, s); break;   // 256 input channels: eight planes of an 8-row patch = 120 KB, one workgroup per CU
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return TRTX_OK;
}
// rows per tile: 16 where the LDS plan leaves two workgroups per CU and the map is tall enough to fill them, else 8
int patch_tile_rows(const ConvArgs& a) {
    const int kc = a.CinK / 32;
    return (kc <= 2 && a.H >= 16) ? 16 : 8;
}
int32_t launch_patch(const ConvArgs& a, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    const int th = patch_tile_rows(a);
    switch (a.bn) {
        case 64: return launch_patch_kc<4>(a, in_bytes, w_bytes, th, s);
        case 80: return launch_patch_kc<5>(a, in_bytes, w_bytes, th, s);
        case 128: return launch_patch_kc<8>(a, in_bytes, w_bytes, th, s);
        default: return TRTX_ERR_UNSUPPORTED;
    }
}
template <int NFRAG, int KC>
void launch_patch_group_mi(const ConvPatchGroupArgs& g, int th, hipStream_t s) {
    if (th == 16) TRTX_LAUNCH((conv_patch_group_f16_kernel<NFRAG, KC, 4>), dim3(g.slot_start[g.n] * 8), dim3(256), 0, s, g);
    else TRTX_LAUNCH((conv_patch_group_f16_kernel<NFRAG, KC, 2>), dim3(g.slot_start[g.n] * 8), dim3(256), 0, s, g);
}
bool patch_possible(const ConvArgs& a);   // (defined with the other tactic predicates below)
// every member a resident-patch layer of ONE instantiation (column-tile width, channel slices, tile rows)?
bool patch_group_possible(const ConvArgs* a, int n) {
    for (int k = 0; k < n; ++k)
        if (!patch_possible(a[k]) || a[k].CinK > 128 || a[k].bn != a[0].bn || a[k].CinK != a[0].CinK || patch_tile_rows(a[k]) != patch_tile_rows(a[0]) || a[k].bn == 128 ||
            (double)a[k].N * a[k].H * a[k].W * a[k].ld_in * 2.0 >= 2.0e9)
            return false;
    return true;
}
int32_t launch_patch_group(const ConvArgs* a, int n, hipStream_t s) {
    ConvPatchGroupArgs g{};
    g.n = n;
    const int th = patch_tile_rows(a[0]);
    int order[kMaxConvGroup];   // same k-steps per tile everywhere: the big maps first
    for (int k = 0; k < n; ++k) order[k] = k;
    std::sort(order, order + n, [&](int x, int y) {
        const long mx = (long)a[x].N * a[x].H * a[x].W, my = (long)a[y].N * a[y].H * a[y].W;
        return mx != my ? mx > my : x < y;
    });
    int slots = 0;
    for (int k = 0; k < n; ++k) {
        const ConvArgs& ak = a[order[k]];
        g.p[k] = ak;
        g.p[k].M = ak.N * ak.Ho * ak.Wo;
        g.tiles_n[k] = ak.Cout_pad / ak.bn;
        g.tiles_x[k] = (ak.W + 15) / 16;
        g.tiles_y[k] = (ak.H + th - 1) / th;
        g.tiles[k] = ak.N * g.tiles_y[k] * g.tiles_x[k] * g.tiles_n[k];
        g.chunk[k] = (g.tiles[k] + 7) / 8;
        g.slot_start[k] = slots;
        slots += g.chunk[k];
        g.in_bytes[k] = (unsigned)((((size_t)ak.N * ak.H * ak.W - 1) * ak.ld_in + ak.Cin) * 2);
        g.w_bytes[k] = (unsigned)((size_t)ak.Cout_pad * ak.Kpad * 2);
    }
    for (int k = n; k <= kMaxConvGroup; ++k) g.slot_start[k] = slots;
    const int kc = a[0].CinK / 32;
#define TRTX_PG(NF)                                           \
    switch (kc) {                                             \
        case 1: launch_patch_group_mi<NF, 1>(g, th, s); break; \
        case 2: launch_patch_group_mi<NF, 2>(g, th, s); break; \
        case 3: launch_patch_group_mi<NF, 3>(g, th, s); break; \
        case 4: launch_patch_group_mi<NF, 4>(g, th, s); break; \
        default: return TRTX_ERR_UNSUPPORTED;                  \
    }
    if (a[0].bn == 64) { TRTX_PG(4) } else { TRTX_PG(5) }
#undef TRTX_PG
    return TRTX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Small-M variant (20x20 maps at batch 32 give M = 12800: 100 tiles of 128 rows for 256 CUs, and a 3x3 conv over 256
// channels is a chain of 72 dependent k-steps).  Here a workgroup owns a 64 x BN tile and its four waves SPLIT K: each
// wave runs its own quarter of the k-steps through a wave-private LDS pipeline (no barrier inside the loop, no coupling
// between waves), the four partial accumulators meet in LDS once at the end.  Twice the tiles, a quarter of the chain.
template <int NFRAG>
__global__ __launch_bounds__(256) void conv_igemm_wsk_f16_kernel(const ConvArgs p, unsigned in_bytes, unsigned w_bytes, int tiles_n,
                                                                 int total_tiles, int xcd_chunk) {
    constexpr int BKT = 32, ROW_B = 64, WM = 64;
    constexpr int BN = 16 * NFRAG;
    constexpr int A_LOADS = WM / 16;                      // 16 rows per wave-instruction
    constexpr int B_LOADS = NFRAG;
    constexpr int A_BYTES = WM * ROW_B;
    constexpr int STAGE_BYTES = A_BYTES + BN * ROW_B;
    constexpr int WAVE_BYTES = NSTAGE * STAGE_BYTES;
    constexpr int LOADS_PER_TILE = A_LOADS + B_LOADS;
    constexpr int PS = BN * 4 + 16;                       // partial-tile row stride (fp32, padded)
    static_assert(WM * PS <= WAVE_BYTES, "partial tile must fit in the wave's stage buffers");
    __shared__ __attribute__((aligned(16))) char smem[4 * WAVE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;
    if (xcd_chunk) {
        tile = (tile & 7) * xcd_chunk + (tile >> 3);
        if (tile >= total_tiles) return;
    }
    const int m0 = (tile / tiles_n) * WM;
    const int n0 = (tile % tiles_n) * BN;
    char* mine = smem + wave * WAVE_BYTES;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, w_bytes, 0x00020000);

    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ swz<32>(lrow);
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = __builtin_amdgcn_rcpf((float)HoWo), inv_wo = __builtin_amdgcn_rcpf((float)p.Wo);  // +-1 estimates, fixed up
    unsigned a_base[A_LOADS], a_rows[A_LOADS], a_cols[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + i * 16 + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        int n = (int)((float)mm * inv_howo);
        int rem = mm - n * HoWo;
        if (rem < 0) { --n; rem += HoWo; }
        if (rem >= HoWo) { ++n; rem -= HoWo; }
        int ho = (int)((float)rem * inv_wo);
        int wo = rem - ho * p.Wo;
        if (wo < 0) { --ho; wo += p.Wo; }
        if (wo >= p.Wo) { ++ho; wo -= p.Wo; }
        const int hi0 = ho * p.stride_h - p.pad_h;
        const int wi0 = wo * p.stride_w - p.pad_w;
        a_base[i] = (unsigned)(((n * p.H + hi0) * p.W + wi0) * p.ld_in + lchunk * 8) * 2u;
        // taps inside the image form a contiguous range (dilation 1): closed form instead of a loop over taps
        a_rows[i] = ok ? tap_range_mask(hi0, p.kh, p.H) : 0u;
        a_cols[i] = tap_range_mask(wi0, p.kw, p.W);
    }
    const int cmax = p.Cin - lchunk * 8;

    // this wave's contiguous share of the k-steps
    const int nk = p.Kpad / BKT;
    const int k_begin = (nk * wave) / 4, k_end = (nk * (wave + 1)) / 4;
    const int steps = k_end - k_begin;
    const int spt = p.CinK / BKT;  // k-steps per filter tap
    int s_kt = k_begin;
    int s_uc, s_r, s_q;
    {
        const int tap = k_begin / spt;
        s_uc = (k_begin - tap * spt) * BKT;
        s_r = tap / p.kw;
        s_q = tap - s_r * p.kw;
    }
    unsigned s_toff = (unsigned)((s_r * p.dil_h * p.W + s_q * p.dil_w) * p.ld_in) * 2u;
    unsigned b_off[B_LOADS];
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j)
        b_off[j] = (unsigned)(((n0 + j * 16 + lrow) * p.Kpad + k_begin * BKT + lchunk * 8) * 2);

    auto issue_tile = [&](int stage) {
        char* sbase = mine + stage * STAGE_BYTES;
        const bool live = s_kt < k_end;
        const unsigned add = s_toff + (unsigned)s_uc * 2u;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const bool ok = ((a_rows[i] >> s_r) & (a_cols[i] >> s_q) & 1u) && s_uc < cmax && live;
            const unsigned voff = ok ? a_base[i] + add : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(sbase + i * 16 * ROW_B), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            const unsigned voff = live ? b_off[j] : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sbase + A_BYTES + j * 16 * ROW_B), 16, voff, 0, 0, 0);
            b_off[j] += BKT * 2;
        }
        ++s_kt;
        s_uc += BKT;
        const int wrap = s_uc >= p.CinK;
        s_uc = wrap ? 0 : s_uc;
        s_q += wrap;
        const int wq = s_q == p.kw;
        s_q = wq ? 0 : s_q;
        s_r += wq;
        s_toff = (unsigned)((s_r * p.dil_h * p.W + s_q * p.dil_w) * p.ld_in) * 2u;
    };

    floatx4 acc[A_LOADS][NFRAG];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int f_off = frow * ROW_B + (((lane >> 4) ^ swz<32>(frow)) * 16);

    auto compute = [&](int stage) {
        const char* sb = mine + stage * STAGE_BYTES;
        half8 af[A_LOADS];
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) af[i] = *reinterpret_cast<const half8*>(sb + i * 16 * ROW_B + f_off);
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            const half8 bf = *reinterpret_cast<const half8*>(sb + A_BYTES + j * 16 * ROW_B + f_off);
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, af[i], acc[i][j], 0, 0, 0);
        }
    };

#define TRTX_WSTEP(S)                                                         \
    {                                                                         \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_TILE) : "memory"); \
        issue_tile(((S) + 2) % NSTAGE);                                       \
        compute(S);                                                           \
    }
    issue_tile(0);
    issue_tile(1);
    for (int kt = 0; kt < steps;) {
        TRTX_WSTEP(0);
        if (++kt >= steps) break;
        TRTX_WSTEP(1);
        if (++kt >= steps) break;
        TRTX_WSTEP(2);
        ++kt;
    }
#undef TRTX_WSTEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- partial tiles -> LDS (wave-private, same bytes the wave's own DMA just finished with), then wave w finishes
    // rows 16w .. 16w+15: sum of the four partials, bias/BN, act1, (+residual, act2), row-major 16-byte stores
    const int px_in = lane & 15;
    const int ch_in = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j)
            *reinterpret_cast<floatx4*>(mine + (i * 16 + px_in) * PS + (j * 16 + ch_in) * 4) = acc[i][j];
    __syncthreads();
    _Float16* __restrict__ out = static_cast<_Float16*>(p.out);
    const _Float16* __restrict__ res = static_cast<const _Float16*>(p.residual);
    const bool second = res || p.act2 != ACT_NONE;
    constexpr int CPR = BN / 8;                  // 8-channel items per row
    constexpr int ITEMS = 16 * CPR;              // per wave
    // rolled, with wave-uniform branches: the finishing code exists once (see the epilogue of conv_igemm_f16_kernel)
#pragma nounroll
    for (int q = lane; q < ITEMS; q += 64) {
        const int row = wave * 16 + q / CPR, cc = q % CPR;
        const int m = m0 + row;
        const int co = n0 + cc * 8;
        if (m >= p.M || co >= p.Cout) continue;
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const char* src = smem + w * WAVE_BYTES + row * PS + cc * 32;
            const floatx4 lo = *reinterpret_cast<const floatx4*>(src);
            const floatx4 hi = *reinterpret_cast<const floatx4*>(src + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] += lo[e];
                x[4 + e] += hi[e];
            }
        }
        if (p.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
            x[0] += b0.x; x[1] += b0.y; x[2] += b0.z; x[3] += b0.w;
            x[4] += b1.x; x[5] += b1.y; x[6] += b1.z; x[7] += b1.w;
        }
        half8 v;
        if (p.act1 == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-x[e])));
        } else if (p.act1 == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e] > 0.f ? x[e] : 0.f);
        } else if (p.act1 == ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = round_to_half(x[e]);
        } else {
#pragma nounroll
            for (int e = 0; e < 8; ++e) v[e] = round_to_half(act_slow(x[e], p.act1, p.alpha1));
        }
        const bool vec = !p.scalar_out;
        if (second) {
            half8 rv = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (res) {
                if (vec) {
                    rv = *reinterpret_cast<const half8*>(res + (size_t)m * p.ld_res + co);
                } else {
#pragma nounroll
                    for (int e = 0; e < 8; ++e)
                        if (co + e < p.Cout) rv[e] = res[(size_t)m * p.ld_res + co + e];
                }
            }
            if (p.act2 == ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = round_to_half((float)v[e] + (float)rv[e]);
            } else if (p.act2 == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = (float)v[e] + (float)rv[e];
                    v[e] = round_to_half(t > 0.f ? t : 0.f);
                }
            } else {
#pragma nounroll
                for (int e = 0; e < 8; ++e) v[e] = round_to_half(act_apply((float)v[e] + (float)rv[e], p.act2, p.alpha2));
            }
        }
        if (vec) {
            *reinterpret_cast<half8*>(out + (size_t)m * p.ld_out + co) = v;
        } else {
#pragma nounroll
            for (int e = 0; e < 8; ++e)
                if (co + e < p.Cout) out[(size_t)m * p.ld_out + co + e] = v[e];
        }
    }
}

template <int NFRAG>
void launch_wsk(const ConvArgs& a, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    const int BN = 16 * NFRAG;
    const int tiles_m = (a.M + 63) / 64, tiles_n = a.Cout_pad / BN;
    const int total = tiles_m * tiles_n;
    const int chunk = (total + 7) / 8;
    TRTX_LAUNCH((conv_igemm_wsk_f16_kernel<NFRAG>), dim3(chunk * 8), dim3(256), 0, s, a, in_bytes, w_bytes, tiles_n, total,
                       chunk);
}

template <int NFRAG, int BKT, int TPS, bool I8 = false, int MI = 2>
void launch(const ConvArgs& a, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    const int BN = 16 * NFRAG, BMT = 64 * MI;
    const int tiles_m = (a.M + BMT - 1) / BMT, tiles_n = a.Cout_pad / BN;
    const int total = tiles_m * tiles_n;
    constexpr bool plain = false;   // (XCD-aware tile order always)
    const int chunk = (total + 7) / 8;
    const int dbg = options().conv_dbg;   // ablation builds only (the product kernels ignore it)
    const bool rs_on = a.t_rs != 0;
    if constexpr (TPS == 1 && !I8) {
        if (a.up_C > 0) {   // folded upsample (conv_igemm_supported has checked the geometry)
            if (rs_on)
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, BKT, TPS, I8, MI, 1, 0, false, 4, true, true>), dim3(plain ? total : chunk * 8), dim3(256), 0, s, a,
                            in_bytes, w_bytes, tiles_n, total, chunk, dbg);
            else
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, BKT, TPS, I8, MI, 1, 0, false, 4, false, true>), dim3(plain ? total : chunk * 8), dim3(256), 0, s, a,
                            in_bytes, w_bytes, tiles_n, total, chunk, dbg);
            return;
        }
    }
    if constexpr (TPS == 1) {   // (int8 too since round 5: the addressing and the operand path do not depend on what the 16 bytes hold)
        if (a.kh == 1 && a.kw == 1 && a.stride_h == 1 && a.stride_w == 1 && a.pad_h == 0 && a.pad_w == 0 && a.Cin == a.CinK && a.Kpad == a.K) {
            if (rs_on)
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, BKT, TPS, I8, MI, 1, 0, false, 4, true, false, true>), dim3(plain ? total : chunk * 8), dim3(256), 0, s, a,
                            in_bytes, w_bytes, tiles_n, total, chunk, dbg);
            else
                TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, BKT, TPS, I8, MI, 1, 0, false, 4, false, false, true>), dim3(plain ? total : chunk * 8), dim3(256), 0, s, a,
                            in_bytes, w_bytes, tiles_n, total, chunk, dbg);
            return;
        }
    }
    if (rs_on)
        TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, BKT, TPS, I8, MI, 1, 0, false, 4, true>), dim3(plain ? total : chunk * 8), dim3(256), 0, s, a, in_bytes,
                    w_bytes, tiles_n, total, chunk, dbg);
    else
    TRTX_LAUNCH((conv_igemm_f16_kernel<NFRAG, BKT, TPS, I8, MI>), dim3(plain ? total : chunk * 8), dim3(256), 0, s, a, in_bytes,
                       w_bytes, tiles_n, total, chunk, dbg);
}

template <int BKT, int TPS, bool I8 = false, int MI = 2>
int32_t launch_bn(const ConvArgs& a, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
    if constexpr (MI == 4) {  // 256-row tiles: instantiated for the column widths of the large-map layers
        switch (a.bn) {
            case 32: launch<2, BKT, TPS, I8, MI>(a, in_bytes, w_bytes, s); break;
            case 64: launch<4, BKT, TPS, I8, MI>(a, in_bytes, w_bytes, s); break;
            case 80: launch<5, BKT, TPS, I8, MI>(a, in_bytes, w_bytes, s); break;
            default: return TRTX_ERR_UNSUPPORTED;
        }
        return TRTX_OK;
    }
    switch (a.bn) {
        case 16: launch<1, BKT, TPS, I8, MI>(a, in_bytes, w_bytes, s); break;
        case 32: launch<2, BKT, TPS, I8, MI>(a, in_bytes, w_bytes, s); break;
        case 64: launch<4, BKT, TPS, I8, MI>(a, in_bytes, w_bytes, s); break;
        case 80: launch<5, BKT, TPS, I8, MI>(a, in_bytes, w_bytes, s); break;
        case 128: launch<8, BKT, TPS, I8, MI>(a, in_bytes, w_bytes, s); break;
        default: return TRTX_ERR_UNSUPPORTED;
    }
    return TRTX_OK;
}

bool valid_bn(int bn) { return bn == 16 || bn == 32 || bn == 64 || bn == 80 || bn == 128; }

// the wave-split-K variant exists for 64- and 80-wide column tiles of fp16 layers with 32-wide k-steps
bool wsk_possible(const ConvArgs& a) {
    if (a.up_C) return false;   // the folded upsample exists in the main kernel only
    return !a.in_i8 && !a.out_i8 && !a.res_i8 && a.bk == 32 && a.CinK % 32 == 0 && (a.bn == 64 || a.bn == 80) && a.Kpad / 32 >= 4;
}
// ... and is what the untuned dispatch picks for few tiles with a long k-chain
bool wsk_default(const ConvArgs& a) {
    const bool no_wsk = !options().wsk;  // A/B switch (TRTX_CONV_NOWSK)
    const int tiles128 = ((a.M + 127) / 128) * (a.Cout_pad / a.bn);
    return !no_wsk && wsk_possible(a) && tiles128 <= 256 && a.Kpad / 32 >= 16;
}
bool r3_possible(const ConvArgs&) { return false; }   // (the row-reuse kernel of rounds 2-4 left the library in round 5: tools/hip/experiments/README.md)
// the resident-patch kernel: fp16 3x3 stride 1 pad 1 over at most 128 (or exactly 256) input channels, 16-byte output stores, 64 / 80 / 128-wide column tiles
bool patch_possible(const ConvArgs& a) {
    return !a.up_C && !a.in_i8 && !a.out_i8 && !a.res_i8 && a.kh == 3 && a.kw == 3 && a.stride_h == 1 && a.stride_w == 1 && a.pad_h == 1 && a.pad_w == 1 &&
           a.dil_h == 1 && a.dil_w == 1 && a.groups == 1 && a.bk == 32 && a.CinK % 32 == 0 && (a.CinK <= 128 || a.CinK == 256) && a.Cin % 8 == 0 && a.Kpad == 9 * a.CinK &&
           !a.scalar_out && a.Ho == a.H && a.Wo == a.W && (a.bn == 64 || a.bn == 80 || a.bn == 128) && a.Cout_pad % a.bn == 0 && (a.bm == 0 || a.bm == 128) &&
           a.t_r3 == 0;
}
// 64-row tiles are instantiated for the fp16 one-tap-per-step kernels (both k-step widths)
bool bm64_possible(const ConvArgs& a) { return !a.in_i8 && a.CinK != 16; }
// ... 256-row tiles too, for 32/64/80-wide column tiles
bool bm256_possible(const ConvArgs& a) { return !a.in_i8 && a.CinK != 16 && (a.bn == 32 || a.bn == 64 || a.bn == 80); }

}  // namespace

static thread_local LaunchProbe* g_launch_probe = nullptr;
void conv_set_launch_probe(LaunchProbe* p) { g_launch_probe = p; }
LaunchProbe* conv_launch_probe() { return g_launch_probe; }

int conv_igemm_pick_bn(int cout) {
    // widest tile that divides the padded Cout without waste
    if (cout % 128 == 0) return 128;
    if (cout % 80 == 0) return 80;
    if (cout % 64 == 0) return 64;
    if (cout % 32 == 0) return 32;
    if (cout % 16 == 0) return 16;
    const int pad16 = (cout + 15) / 16 * 16;
    return conv_igemm_pick_bn(pad16);
}

int conv_igemm_pick_bk(int cin, int taps) {
    // 64-wide steps touch whole 128-B lines (the vector L1 serves lines, not halves) and halve the barriers, but their
    // 24..32 KB stages are double- instead of triple-buffered and occupancy drops.  Measured on YOLOv8n b32: isolated
    // 3x3 layers over Cin % 64 == 0 gain 5-12 %, 1x1 layers lose 15-20 %, and with 3x3-only selection the whole engine
    // step is still 4 % slower (1.61 vs 1.54 ms on the same box): not a static default; the tactic tuner (runtime/tune.cpp)
    // times it per layer (the packed weights of a Cin % 64 == 0 layer are the same for both widths).
    (void)cin; (void)taps;
    return 32;
}

int conv_igemm_pick_cink(int cin, int bk) {
    if (cin <= 16 && bk == 32) return 16;  // two taps per k-step
    return (cin + bk - 1) / bk * bk;
}

bool conv_igemm_supported(const ConvArgs& a) {
    if (a.bn == 256) return conv_gemm256_possible(a) && a.t_r3 == 0 && a.Kpad == (a.kh * a.kw * a.CinK + 63) / 64 * 64;   // the 256 x 256 x 64 tactic
    if (a.in_i8 && (a.bk != 32 || a.CinK % 32 || a.scalar_out)) return false;  // int8: 64-channel k-steps, vector epilogue
    if (a.up_C != 0 && (a.up_C < 0 || a.in_i8 || a.kh != 1 || a.kw != 1 || a.stride_h != 1 || a.stride_w != 1 || a.pad_h || a.pad_w || a.up_C % 64 || a.up_C >= a.Cin ||
                        a.H != 2 * a.up_H || a.W != 2 * a.up_W || a.up_ld % 8 || a.CinK == 16 || a.t_r3 != 0))
        return false;  // folded upsample: 1x1 stride 1, a whole number of k-steps from the half-resolution tensor
    if ((a.out_i8 || a.res_i8) && a.scalar_out) return false;
    const bool out_vec = a.ld_out % 8 == 0 && a.Cout % 8 == 0 && (!a.residual || a.ld_res % 8 == 0);
    const int bk = a.CinK % 64 == 0 && a.bk == 64 ? 64 : 32;
    const bool cink_ok = a.CinK % bk == 0 || (a.CinK == 16 && bk == 32);
    const double img_bytes = (double)a.H * a.W * a.ld_in * 2.0, w_b = (double)a.Cout_pad * a.Kpad * 2.0;
    return a.Cin % 8 == 0 && a.ld_in % 8 == 0 && a.groups == 1 && a.dil_h == 1 && a.dil_w == 1 && a.kh * a.kw <= kMaxTaps && cink_ok && a.CinK >= a.Cin &&
           a.Kpad == (a.kh * a.kw * a.CinK + bk - 1) / bk * bk && (out_vec || a.scalar_out) && img_bytes < 2.0e9 && w_b < 2.0e9 &&
           valid_bn(a.bn) && a.Cout_pad % a.bn == 0 && (a.bm == 0 || a.bm == 128 || (a.bm == 64 && bm64_possible(a)) || (a.bm == 256 && bm256_possible(a))) &&
           (a.t_r3 == 0 || r3_possible(a))  && a.t_ws != 6;   // (wave roles, t_ws == 6, are an fp32 tactic: measured on the fp16 tiles, -3..-17 % on small maps alone, +-0 on the bench line: profiles/r05_fp16_roles_*)
}

// ---- tactics: the launch configurations of one layer that produce the SAME packed-weight layout, so that they can be exchanged
// at run time.  Entry 0 is what the untuned dispatch does.  Tiles of any width / height accumulate every output element over
// K in the same order (bit-identical results); the wave-split-K and weight-stationary kernels sum in a different order (fp16
// results may differ in the last place).
int conv_tactics(const ConvArgs& a0, ConvTactic* out, int max_out, bool work_efficient_only) {
    int n = 0;
    auto push = [&](int bn, int bk, int bm, int wsk, int ws, int r3 = 0) {
        for (int i = 0; i < n; ++i)
            if (out[i].bn == bn && out[i].bk == bk && out[i].bm == bm && out[i].wsk == wsk && out[i].ws == ws && out[i].r3 == r3) return;
        if (n < max_out) out[n++] = ConvTactic{bn, bk, bm, wsk, ws, r3};
    };
    ConvArgs a = a0;
    a.bm = 0; a.t_wsk = 0; a.t_ws = 0; a.t_r3 = 0;
    if (!conv_igemm_supported(a)) return 0;
    const bool fp16 = !a.in_i8 && !a.out_i8 && !a.res_i8;
    const bool pinned = a0.k_pinned != 0;   // one summation order whatever the tuner measures: no wave-split-K, no weight-stationary candidate
    const bool ws_ok = fp16 && !pinned && conv_ws_supported(a);
    // 0: the default.  Work-efficient sets (engines whose contexts share the chip) never split K over the waves, not even as the
    // starting point: measured on YOLOv8n b32 with three contexts in flight it is worth nothing there (33.0k img/s with, 33.1k without,
    // two runs each on one box), and "one summation order per plan" is the simpler contract.
    if (ws_ok) push(a.bn, a.bk, 128, 1, 2);
    else push(a.bn, a.bk, 128, (wsk_default(a) && !work_efficient_only && !pinned) ? 2 : 1, 1);
    // the 256 x 256 x 64 role-alternating tile for large plain GEMMs (conv_gemm256.hip): more work-efficient than any 128-row tile
    if (fp16 && options().gemm256 && conv_gemm256_worthwhile(a)) push(256, 64, 256, 1, 1);
    const int bks[2] = {a.bk, (fp16 && a.CinK % 64 == 0 && a.CinK != 16 && a.Kpad % 64 == 0) ? (a.bk == 32 ? 64 : 32) : a.bk};
    static const int bns[5] = {128, 80, 64, 32, 16};
    for (int bi = 0; bi < 5; ++bi) {
        const int bn = bns[bi];
        if (a.Cout_pad % bn) continue;
        if (bn == 16 && a.Cout_pad > 32 && bn != a.bn) continue;  // 16-wide tiles re-read the A tile Cout/16 times: only for tiny Cout
        if (work_efficient_only && bn != a.bn && bn != 64) continue;  // 64-wide tiles stay: they let a whole network share one kernel
        for (int ki = 0; ki < 2; ++ki) {
            ConvArgs t = a;
            t.bn = bn;
            t.bk = bks[ki];
            if (ki == 1 && bks[1] == bks[0]) continue;
            push(bn, t.bk, 128, 1, 1);
            if (bm64_possible(t) && !work_efficient_only) push(bn, t.bk, 64, 1, 1);
            if (bm256_possible(t) && (long)((a.M + 255) / 256) * (a.Cout_pad / bn) >= 512) push(bn, t.bk, 256, 1, 1);  // >= 2 tiles per CU
            if (wsk_possible(t) && !work_efficient_only && !pinned) push(bn, t.bk, 128, 2, 1);
            // the resident-patch 3x3 kernel (ws == 3): the same bits, fewer bytes through the global -> LDS fill path (work-efficient: a candidate in every set)
            if (fp16 && options().patch && patch_possible(t)) push(bn, t.bk, 128, 1, 3);
            // the resident-operand kernels (ws == 7 / 8, conv_res.hip): the same bits again, nothing fetched inside the k-loop.  Measured on YOLOv8n b32
            // (profiles/r06_engine_ab.txt, same box, alternating): the 3x3 kernel alone - one context 1.208 -> 1.188 ms, three contexts in flight 37.7-38.2k ->
            // 37.9k img/s; with the 1x1 kernel too - one context 1.174 ms, three contexts 37.1-37.2k: its 16-wave persistent workgroups leave another context's
            // kernels less room on a CU than they gain, so engines built for contexts in flight (work_efficient_only) list the 3x3 kernel only
            if (fp16 && (options().res & 1) && conv_res_possible(t)) push(bn, t.bk, 128, 1, 7);
            if (fp16 && (options().res & 2) && !work_efficient_only && conv_res1_possible(t)) push(bn, t.bk, 128, 1, 8);   // ... and its 1x1 sibling (ws == 8)
        }
    }
    return n;
}

void conv_apply_tactic(ConvArgs* a, const ConvTactic& t) {
    a->bn = t.bn;
    a->bk = t.bk;
    a->bm = t.bm;
    a->t_wsk = t.wsk;
    a->t_ws = t.ws;
    a->t_r3 = t.r3;
}

int32_t conv_igemm_f16(const ConvArgs& a0, hipStream_t s) {
    if (!conv_igemm_supported(a0) || (a0.in_i8 && !a0.cscale)) return TRTX_ERR_UNSUPPORTED;
    const bool fp16 = !a0.in_i8 && !a0.out_i8 && !a0.res_i8;
    if (a0.bn == 256) return conv_gemm256_f16(a0, s);
    if (fp16 && a0.t_ws == 7 && conv_res_possible(a0)) return conv_res_f16(&a0, 1, s);
    if (fp16 && a0.t_ws == 8 && conv_res1_possible(a0)) return conv_res1_f16(a0, s);
    // small-channel 3x3 / 1x1 fp16 layers: weight-stationary persistent kernel (t_ws: 0 = where supported, 1 = never, 2 = asked for)
    if (fp16 && a0.t_ws != 1 && a0.t_ws != 3 && a0.t_ws != 7 && a0.t_ws != 8 && conv_ws_supported(a0)) return conv_ws_f16(a0, s);   // (3 = the resident-patch kernel, below)
    // The buffer descriptor addresses 32-bit byte offsets: launch over groups of images whose slice stays below 2 GB.
    const size_t img_in = (size_t)a0.H * a0.W * a0.ld_in * 2;
    const int per = (int)std::max<size_t>(1, (size_t)2000000000 / img_in);
    const unsigned w_bytes = (unsigned)((size_t)a0.Cout_pad * a0.Kpad * 2);
    for (int n0 = 0; n0 < a0.N; n0 += per) {
        ConvArgs a = a0;
        a.N = std::min(per, a0.N - n0);
        a.M = a.N * a.Ho * a.Wo;
        a.in = static_cast<const char*>(a0.in) + (size_t)n0 * img_in;
        a.out = static_cast<char*>(a0.out) + (size_t)n0 * a.Ho * a.Wo * a.ld_out * (a0.out_i8 ? 1 : 2);
        if (a0.residual) a.residual = static_cast<const char*>(a0.residual) + (size_t)n0 * a.Ho * a.Wo * a.ld_res * (a0.res_i8 ? 1 : 2);
        if (a0.up_C) a.up_in = static_cast<const char*>(a0.up_in) + (size_t)n0 * a.up_H * a.up_W * a.up_ld * 2;
        // extent of the addressed slice: last pixel's first byte + the channels this conv reads
        const unsigned in_bytes = (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ld_in + a.Cin) * 2);
        // few tiles and a long k-chain: the wave-split-K variant (see conv_igemm_wsk_f16_kernel); t_wsk: 0 = that rule, 1 = never,
        // 2 = wherever the variant exists
        const bool wsk = a.t_wsk == 1 ? false : (a.t_wsk == 2 ? wsk_possible(a) : wsk_default(a));
        const bool bm64 = a.bm == 64, bm256 = a.bm == 256;
        int32_t st = TRTX_OK;
        if (a.in_i8) {
            st = launch_bn<32, 1, true>(a, in_bytes, w_bytes, s);
        } else if (a.t_ws == 3 && patch_possible(a)) {
            st = launch_patch(a, in_bytes, w_bytes, s);
        } else if (wsk) {
            if (a.bn == 64) launch_wsk<4>(a, in_bytes, w_bytes, s);
            else launch_wsk<5>(a, in_bytes, w_bytes, s);
        } else if (a.bk == 64) {
            st = bm64 ? launch_bn<64, 1, false, 1>(a, in_bytes, w_bytes, s)
                      : (bm256 ? launch_bn<64, 1, false, 4>(a, in_bytes, w_bytes, s) : launch_bn<64, 1>(a, in_bytes, w_bytes, s));
        } else if (a.CinK == 16) {
            st = launch_bn<32, 2>(a, in_bytes, w_bytes, s);
        } else {
            st = bm64 ? launch_bn<32, 1, false, 1>(a, in_bytes, w_bytes, s)
                      : (bm256 ? launch_bn<32, 1, false, 4>(a, in_bytes, w_bytes, s) : launch_bn<32, 1>(a, in_bytes, w_bytes, s));
        }
        if (st != TRTX_OK) return st;
    }
    return check_launch("conv_igemm_f16");
}

// ---- grouped launch ------------------------------------------------------------------------------------------------------
namespace {
bool plain_gemm(const ConvArgs& a) {   // the ONE instantiation's condition (launch<> above)
    return a.kh == 1 && a.kw == 1 && a.stride_h == 1 && a.stride_w == 1 && a.pad_h == 0 && a.pad_w == 0 && a.Cin == a.CinK && a.Kpad == a.K;
}
// what the single-problem dispatch would run for this layer must be the plain 128-row / 32-wide-step main kernel
bool group_member_ok(const ConvArgs& a) {
    return conv_igemm_supported(a) && (a.in_i8 || (!a.out_i8 && !a.res_i8)) && !a.up_C && !a.scalar_out && a.CinK != 16 && a.bk == 32 && (a.bn == 64 || a.bn == 80) &&
           (a.bm == 0 || a.bm == 128) && a.t_r3 == 0 && a.t_wsk != 2 && a.t_ws != 3 && (double)a.N * a.H * a.W * a.ld_in * 2.0 < 2.0e9;
}
template <int NFRAG, bool RS, bool ONE, bool I8 = false>
void launch_group(const ConvGroupArgs& g, hipStream_t s) {
    const int dbg = options().conv_dbg;
    TRTX_LAUNCH((conv_igemm_group_f16_kernel<NFRAG, 32, RS, ONE, I8>), dim3(g.slot_start[g.n] * 8), dim3(256), 0, s, g, dbg);
}
}  // namespace

bool conv_igemm_group_supported(const ConvArgs* a, int n) {
    if (n < 2 || n > kMaxConvGroup) return false;
    for (int k = 0; k < n; ++k) {
        if (!group_member_ok(a[k])) return false;
        if (a[k].t_ws != 1 && conv_ws_supported(a[k])) return false;   // that layer belongs to the weight-stationary kernel
        if (a[k].bn != a[0].bn || a[k].in_i8 != a[0].in_i8 || (a[k].t_rs != 0) != (a[0].t_rs != 0) || plain_gemm(a[k]) != plain_gemm(a[0])) return false;
    }
    return true;
}

int32_t conv_igemm_group_f16(const ConvArgs* a, int n, hipStream_t s) {
    if (!conv_igemm_group_supported(a, n)) return TRTX_ERR_UNSUPPORTED;
    // groups whose members are all resident-patch layers of one instantiation run on that kernel (round 5: +1 % on the bench line over the main kernel's
    // groups, same bits - profiles/r05_patch_r3_first_run.txt; groups are not tuned, TRTX_CONV_PATCH=0 is the A/B switch)
    if ((options().res & 4) && conv_res_group_possible(a, n)) return conv_res_f16(a, n, s);   // (the detect head's second 3x3 of every level: one context 1.196 -> 1.188 ms)
    if (options().patch && patch_group_possible(a, n)) {
        const int32_t st = launch_patch_group(a, n, s);
        return st != TRTX_OK ? st : check_launch("conv_patch_group_f16");
    }
    ConvGroupArgs g{};
    g.n = n;
    int order[kMaxConvGroup];   // falling k-steps per tile (ties: more rows first)
    for (int k = 0; k < n; ++k) order[k] = k;
    std::sort(order, order + n, [&](int x, int y) {
        if (a[x].Kpad != a[y].Kpad) return a[x].Kpad > a[y].Kpad;
        const long mx = (long)a[x].N * a[x].Ho * a[x].Wo, my = (long)a[y].N * a[y].Ho * a[y].Wo;
        return mx != my ? mx > my : x < y;
    });
    int slots = 0;
    for (int k = 0; k < n; ++k) {
        const ConvArgs& ak = a[order[k]];
        g.p[k] = ak;
        g.p[k].M = ak.N * ak.Ho * ak.Wo;
        g.tiles_n[k] = ak.Cout_pad / ak.bn;
        g.tiles[k] = ((g.p[k].M + 127) / 128) * g.tiles_n[k];
        g.chunk[k] = (g.tiles[k] + 7) / 8;
        g.slot_start[k] = slots;
        slots += g.chunk[k];
        g.in_bytes[k] = (unsigned)((((size_t)ak.N * ak.H * ak.W - 1) * ak.ld_in + ak.Cin) * 2);
        g.w_bytes[k] = (unsigned)((size_t)ak.Cout_pad * ak.Kpad * 2);
    }
    for (int k = n; k <= kMaxConvGroup; ++k) g.slot_start[k] = slots;
    const bool rs = a[0].t_rs != 0;
    const bool one = plain_gemm(a[0]);
    const int nf = a[0].bn / 16;
    if (a[0].in_i8) {
        for (int k = 0; k < n; ++k)
            if (!a[k].cscale) return TRTX_ERR_UNSUPPORTED;
        if (nf == 4) {
            if (one) rs ? launch_group<4, true, true, true>(g, s) : launch_group<4, false, true, true>(g, s);
            else rs ? launch_group<4, true, false, true>(g, s) : launch_group<4, false, false, true>(g, s);
        } else {
            if (one) rs ? launch_group<5, true, true, true>(g, s) : launch_group<5, false, true, true>(g, s);
            else rs ? launch_group<5, true, false, true>(g, s) : launch_group<5, false, false, true>(g, s);
        }
    } else if (nf == 4) {
        if (one) rs ? launch_group<4, true, true>(g, s) : launch_group<4, false, true>(g, s);
        else rs ? launch_group<4, true, false>(g, s) : launch_group<4, false, false>(g, s);
    } else {
        if (one) rs ? launch_group<5, true, true>(g, s) : launch_group<5, false, true>(g, s);
        else rs ? launch_group<5, true, false>(g, s) : launch_group<5, false, false>(g, s);
    }
    return check_launch("conv_igemm_group_f16");
}

}  // namespace trtx
