// Resident-operand 3x3 convolution for gfx950 (MI355X), tactic ConvArgs::t_ws == 7 (round 6).  The kernel behind the 3x3 stride-1 layers of YOLOv8n's C2f
// bottlenecks and detect head (yolov8/src/block.cpp:79-155, model.cpp:188-251) whose whole weight slab fits in LDS: 32 -> 32, 64 -> 64, 64 -> 80.
//
// Why (profiles/r05_igemm_f16_residency.txt, r04_lds_fill_model.txt): the implicit-GEMM kernels re-fill the weight tile of every k-step of every output tile
// through the global -> LDS path (1.99 GB per YOLOv8n b32 step for 6.3 MB of weights), every k-step costs each wave 3 LDS-DMA pieces of 60-185 issue cycles
// next to 128 cycles of MFMA plus a workgroup barrier, and a lone 128 x 64 tile lives 9.8 us for 1 us of MFMA work.  Here NOTHING is fetched inside the
// k-loop:
//   * PERSISTENT workgroups (one or two per CU): the layer's weights - all 9 * CinK x Cout of them, conv_igemm's swizzled 64-byte rows - are brought to LDS
//     ONCE per workgroup and stay;
//   * the workgroup is 8 waves = two HALVES of 4.  A half owns an output tile of TH x 16 pixels of one image (patch_index.h's geometry: its (TH + 2) x 18
//     input patch lies in LDS, one plane per 32-channel slice) and the halves ALTERNATE ROLES phase by phase: while one half runs its k-loop - ds_read_b128
//     fragments at immediate offsets and MFMAs, fully unrolled, no barrier, no wait, no address arithmetic - the other finishes its previous tile from the
//     accumulator registers (bias, activation, shortcut, stores) and issues the LDS-DMA of its next patch.  One s_barrier per phase.  On every SIMD one wave
//     multiplies while its partner does VALU / memory work: the matrix pipe and the vector ALU are separate pipes (MI355X_MICROARCH.md "Two waves per SIMD");
//   * the EPILOGUE stays in registers.  A lane's accumulator fragment is 4 channels of one pixel; the weight rows of a 16-channel fragment are read from LDS
//     in the order {0-3, 8-11, 4-7, 12-15} so that one v_permlane32_swap per packed dword leaves every lane with 8 CONSECUTIVE channels of its pixel: 16-byte
//     NHWC stores (64 contiguous bytes per pixel and instruction), 16-byte shortcut loads, no LDS staging tile, no barrier (cdna_hip_programming.md T21).
// Every output element accumulates the same products in the same order as in conv_igemm_tile (taps outer, 32-channel slices inner, one
// v_mfma_f32_16x16x32_f16 per step with the same lane -> k mapping) and is finished by the same arithmetic as conv_epilogue_fast: the results are
// BIT-IDENTICAL to the main kernel's (tests/test_gpu_conv.py::test_every_conv_tactic_is_the_same_convolution).
// Several independent layers of one instantiation can share a launch (the detect head's siblings): a workgroup is bound to one problem for its whole life.
#include "../options.h"
#include "igemm_tile.h"

#ifndef TRTX_RES_ABLATE  // the probe's ablation word (1 no stores, 2 no patch DMA after the prologue, 4 no MFMAs, 8 no activation); the constant 0 in the product
#define TRTX_RES_ABLATE 0
#endif
#ifndef TRTX_RES_STAMP   // phase anatomy (tools/hip/res3_anatomy.hip): 0 phase entry, 1 k-loop done / patch DMA issued, 2 epilogue done, 3 waited, 4 past the barrier
#define TRTX_RES_STAMP(ph, i)
#endif

namespace trtx {
namespace {
namespace px = patchidx;

constexpr int kResMax = kMaxConvGroup;   // problems per launch
struct ConvResArgs {
    int n;
    int slot_start[kResMax + 1];   // per XCD: first workgroup slot of problem k (slot_start[n...] = slots per XCD)
    int tiles[kResMax], chunk[kResMax], per[kResMax], tiles_x[kResMax], tiles_y[kResMax];   // chunk = tiles per XCD, per = tiles per workgroup (a contiguous run)
    int tiles_n[kResMax];          // column tiles of the problem (a workgroup is bound to one: slot % tiles_n)
    unsigned in_bytes[kResMax], w_bytes[kResMax];
    ConvArgs p[kResMax];
};

template <int NFRAG, int KC, int MI>
constexpr int res3_lds_bytes() {
    return 9 * KC * 16 * NFRAG * 64 + 2 * KC * (4 * MI + 2) * px::kPitch * px::kPixelBytes;
}

// MFMA row rho of a 16-row weight fragment holds output channel sigma16(rho): lane group g = rho >> 2 then owns channels {0, 8, 4, 12}[g] + [0, 4)
__device__ __forceinline__ int sigma16(int r) { return (r & 3) | ((r & 4) << 1) | ((r & 8) >> 1); }

__device__ __forceinline__ unsigned pack_h2(_Float16 lo, _Float16 hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, v);
}

// The fields of a problem's ConvArgs the kernel uses after its set-up, held in scalar registers.  g.p[pid] is indexed at run time: left to itself the compiler
// re-reads such fields from the kernel-argument segment wherever it wants them (14 s_load + s_waitcnt lgkmcnt(0) inside the phase loop of the first build,
// ~200 cycles each in the dependency chain of the epilogue: 4.2k cycles for 32 outputs per lane, profiles/r06_res3_anatomy_first.txt).  The empty asm makes
// each value opaque - computed once, not re-loadable.
// (global address space spelled out: a pointer that went through the asm is otherwise a generic one - FLAT stores, which count on lgkmcnt as well and make
// every LDS wait of the k-loop a full lgkmcnt(0))
typedef __attribute__((address_space(1))) _Float16 g_half;
typedef __attribute__((address_space(1))) const _Float16 g_chalf;
typedef __attribute__((address_space(1))) const float g_cfloat;
typedef __attribute__((address_space(1))) float g_float;
struct ResP {
    int H, W, Cin, ld_in, Cout, ld_out, ld_res, act1, act2;
    float alpha1, alpha2;   // (fp32 launches: the slow activations' parameter)
    g_cfloat* bias;
    g_half* out;            // fp32 launches: the same addresses seen as g_float (outf() / resf())
    g_chalf* res;
    __device__ __forceinline__ g_float* outf() const { return (g_float*)out; }
    __device__ __forceinline__ g_cfloat* resf() const { return (g_cfloat*)res; }
};
__device__ __forceinline__ int pin_s(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
template <typename P>
__device__ __forceinline__ P* pin_p(P* v) {
    asm volatile("" : "+s"(v));
    return v;
}

// Row slab i of a wave's tile (16 pixels x 16 NFRAG channels) from the accumulators to memory: conv_epilogue_fast's arithmetic, element for element.
// STAGED: every step of the activation runs over all 4 NFRAG values of the lane before the next step starts (sched_barrier between the stages).  Written
// fragment by fragment the compiler interleaved two dependent chains (mul -> exp -> add -> rcp -> mul, quarter-rate transcendentals with their latency) and the
// lone wave of a SIMD issued one instruction per ~10 cycles: 3.0k cycles for 32 outputs per lane with nothing else running (profiles/r06_res3_anatomy.txt).
// `rv` / `rl`: the shortcut's values for this lane's stores (fetched by the caller a phase ahead), zeros without a shortcut.
template <int NFRAG, int ACT1>
__device__ __forceinline__ void res_epilogue_slab(const ResP& p, const floatx4 (&acc)[NFRAG], const floatx4 (&bias)[NFRAG], const half8 (&rv)[NFRAG / 2 > 0 ? NFRAG / 2 : 1],
                                                  half4 rl, bool okpix, int m, int lane) {
    constexpr int NP = NFRAG / 2;             // fragment pairs -> 16-byte stores; an odd last fragment goes out in 8-byte pieces
    constexpr bool ODD = (NFRAG & 1) != 0;
    constexpr int NV = 4 * NFRAG;
    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
    typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
    const bool has_res = p.res != nullptr, relu2 = p.act2 != ACT_NONE;
    const bool second = has_res || relu2;
    const int g = lane >> 4;
    const int cpair = 16 * (lane >> 5) + 8 * (g & 1);            // + 32 jp: first of this lane's 8 channels after the swap
    const int clast = (NFRAG - 1) * 16 + (((g & 1) << 3) | ((g & 2) << 1));   // first of its 4 channels of the unpaired fragment
    _Float16 h[NV];
    // (four fragments = 16 values per lane at a time: enough independent chains for the ALU, half the temporaries of a 128-wide tile)
#pragma unroll
    for (int j0 = 0; j0 < NFRAG; j0 += 4) {
        constexpr int GV = 16;
        float v[GV], t[GV];
#pragma unroll
        for (int k = 0; k < GV; ++k) {
            const int j = j0 + k / 4;
            v[k] = j < NFRAG ? acc[j < NFRAG ? j : 0][k & 3] + bias[j < NFRAG ? j : 0][k & 3] : 0.f;   // (no bias: the registers hold -0.0f, the additive identity of every float)
        }
        if constexpr (ACT1 == ACT_SILU) {
            if (TRTX_RES_ABLATE & 8) {
#pragma unroll
                for (int k = 0; k < GV; ++k) t[k] = 1.0f;
            } else {
#pragma unroll
                for (int k = 0; k < GV; ++k) t[k] = __expf(-v[k]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < GV; ++k) t[k] = 1.0f + t[k];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < GV; ++k) t[k] = __builtin_amdgcn_rcpf(t[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < GV; ++k)
                if (j0 * 4 + k < NV) h[j0 * 4 + k] = round_to_half(v[k] * t[k]);
        } else if constexpr (ACT1 == ACT_RELU) {
#pragma unroll
            for (int k = 0; k < GV; ++k)
                if (j0 * 4 + k < NV) h[j0 * 4 + k] = round_to_half(v[k] > 0.f ? v[k] : 0.f);
        } else {
#pragma unroll
            for (int k = 0; k < GV; ++k)
                if (j0 * 4 + k < NV) h[j0 * 4 + k] = round_to_half(v[k]);
        }
    }
    unsigned pk[NFRAG][2];
#pragma unroll
    for (int j = 0; j < NFRAG; ++j) {
        pk[j][0] = pack_h2(h[4 * j], h[4 * j + 1]);
        pk[j][1] = pack_h2(h[4 * j + 2], h[4 * j + 3]);
    }
    g_half* orow = p.out + (size_t)m * p.ld_out;
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
        // lanes 32-63 of fragment 2jp <-> lanes 0-31 of fragment 2jp + 1: afterwards (a0, a1, b0, b1) are 8 consecutive channels in every lane
        const uintx2 s0 = __builtin_amdgcn_permlane32_swap(pk[2 * jp][0], pk[2 * jp + 1][0], false, false);
        const uintx2 s1 = __builtin_amdgcn_permlane32_swap(pk[2 * jp][1], pk[2 * jp + 1][1], false, false);
        const uintx4 q = {s0[0], s1[0], s0[1], s1[1]};
        half8 o = __builtin_bit_cast(half8, q);
        if (second) {
            if (!relu2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = round_to_half((float)o[e] + (float)rv[jp][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t2 = (float)o[e] + (float)rv[jp][e];
                    o[e] = round_to_half(t2 > 0.f ? t2 : 0.f);
                }
            }
        }
        const int co = 32 * jp + cpair;
        if (okpix & (co < p.Cout) & !(TRTX_RES_ABLATE & 1)) *reinterpret_cast<__attribute__((address_space(1))) half8*>(orow + co) = o;
    }
    if constexpr (ODD) {
        const uintx2 q = {pk[NFRAG - 1][0], pk[NFRAG - 1][1]};
        half4 o = __builtin_bit_cast(half4, q);
        if (second) {
            if (!relu2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = round_to_half((float)o[e] + (float)rl[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t2 = (float)o[e] + (float)rl[e];
                    o[e] = round_to_half(t2 > 0.f ? t2 : 0.f);
                }
            }
        }
        if (okpix & (clast < p.Cout)) *reinterpret_cast<__attribute__((address_space(1))) half4*>(orow + clast) = o;
    }
}

// The fp32 form (conv_epilogue_f32, element for element): the sums started at the bias; act1, (+ shortcut), act2, 16-byte stores straight from the fragment.
// The common activations run STAGED over the slab's 4 NFRAG values (act_f32's operations, one kind at a time): called per element - a chain of scalar
// branches on the activation kind in front of five dependent instructions, and a call site for the rare kinds - it cost 800-950 cycles per output, 12k cycles
// for a 16-output slab against the 9-11k of the tile's k-loop (profiles/r06_res3_f32_anatomy.txt).
template <int NFRAG>
__device__ __forceinline__ void res_epilogue_slab_f32(const ResP& p, const floatx4 (&acc)[NFRAG], const floatx4 (&rres)[NFRAG], bool okpix, int m, int lane) {
    const int grp = lane >> 4;
    g_float* orow = p.outf() + (size_t)m * p.ld_out;
    constexpr int NV = 4 * NFRAG;
    float v[NV];
#pragma unroll
    for (int j = 0; j < NFRAG; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[j][e];
    const bool fast = (p.act1 == ACT_NONE || p.act1 == ACT_RELU || p.act1 == ACT_SILU) && (p.act2 == ACT_NONE || p.act2 == ACT_RELU);
    if (fast) {
        if (p.act1 == ACT_SILU) {
            float t[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) t[k] = __builtin_amdgcn_exp2f(v[k] * -1.44269504088896341f);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NV; ++k) t[k] = 1.0f + t[k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NV; ++k) t[k] = __builtin_amdgcn_rcpf(t[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = v[k] * t[k];
        } else if (p.act1 == ACT_RELU) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        }
        if (p.res) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] += rres[k >> 2][k & 3];
        }
        if (p.act2 == ACT_RELU) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        }
    } else {
        // (the shortcut is added in a loop of its own, unrolled: indexed by the rolled loops' counter the shortcut registers become a scratch buffer)
#pragma unroll 1
        for (int k = 0; k < NV; ++k) v[k] = act_f32(v[k], p.act1, p.alpha1);
        if (p.res) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] += rres[k >> 2][k & 3];
        }
        if (p.act2 != ACT_NONE) {
#pragma unroll 1
            for (int k = 0; k < NV; ++k) v[k] = act_f32(v[k], p.act2, p.alpha2);
        }
    }
#pragma unroll
    for (int j = 0; j < NFRAG; ++j) {
        const int co = 16 * j + 4 * grp;
        const floatx4 xv = {v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
        if (okpix & (co < p.Cout) & !(TRTX_RES_ABLATE & 1)) *reinterpret_cast<__attribute__((address_space(1))) floatx4*>(orow + co) = xv;
    }
}

// NG = wave groups of 4 (2 or 3); PIPE = fragment reads one k-step ahead of their MFMAs (a second register set); WPS = waves per SIMD the launch wants resident
// F32: fp32 operands (the launcher states the input side in 2-byte units, conv_igemm_f32.hip's convention: a plane = 16 floats of every patch pixel, a k-step = 16
// channels = the same 64-byte rows) on v_mfma_f32_16x16x4_f32 with conv_igemm_tile<..., F32>'s two-level K sum (a k-step's 16 products sum from zero, the step's
// partial joins the running total - which starts at the bias - in step order) and conv_epilogue_f32's arithmetic: that kernel's bits.  An fp32 tile multiplies
// 16x longer than an fp16 one while its epilogue and patch fetch cost the same: the rotation keeps the matrix pipe busy but for the barrier between phases.
template <int NFRAG, int KC, int MI, int NG, bool PIPE, int WPS, bool F32 = false>
__global__ __launch_bounds__(NG * 256, WPS) void conv_res3_f16_kernel(const ConvResArgs g) {
    constexpr int BN = 16 * NFRAG;
    constexpr int TH = 4 * MI;
    constexpr int NK = 9 * KC;
    constexpr int PLANE = (TH + 2) * px::kPitch * px::kPixelBytes;
    constexpr int PIECES = PLANE / 1024;
    constexpr int PPW = (PIECES + 3) / 4;      // pieces of a plane per wave of a group
    constexpr int BSTEP = BN * 64;             // the weight rows of one k-step
    constexpr int B_BYTES = NK * BSTEP;
    constexpr int B_PIECES = B_BYTES / 1024;   // 16 rows each
    constexpr int PATCH_BYTES = KC * PLANE;
    constexpr int NW = NG * 4;
    static_assert(NG == 2 || NG == 3, "two or three wave groups");
    static_assert(PLANE % 1024 == 0, "a plane is a whole number of DMA pieces");
    static_assert(B_BYTES + 2 * PATCH_BYTES == res3_lds_bytes<NFRAG, KC, MI>(), "LDS plan");
    __shared__ __attribute__((aligned(16))) char smem[res3_lds_bytes<NFRAG, KC, MI>()];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp_id = wave_all >> 2, wave = wave_all & 3;

    // ---- which problem, which tiles: workgroup id -> (xcd, slot); a problem owns a contiguous range of slots in every XCD and every XCD a contiguous chunk of
    // the problem's tiles (halo rows and the weights stay in one L2); the slot's workgroup takes tiles first, first + nslots, ... of the chunk
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int pid = 0;
#pragma unroll
    for (int k = 1; k < kResMax; ++k) pid += (k < g.n && slot >= g.slot_start[k]) ? 1 : 0;
    pid = __builtin_amdgcn_readfirstlane(pid);
    const ConvArgs& pa = g.p[pid];
    const int per = g.per[pid];
    const int tiles_n = g.tiles_n[pid];
    const int slot_l = slot - g.slot_start[pid];
    const int n0 = pin_s((slot_l % tiles_n) * BN);   // this workgroup's column tile
    const int t_end = min(xcd * g.chunk[pid] + g.chunk[pid], g.tiles[pid]);
    const int first = xcd * g.chunk[pid] + (slot_l / tiles_n) * per;
    const int T = pin_s(first < t_end ? min(per, t_end - first) : 0);   // tiles of this workgroup (wave-uniform): first .. first + T - 1
    if (T == 0) return;
    // A workgroup's tiles are CONSECUTIVE (x fastest, then y, then the image): the coordinates of the next tile follow from the last by two compares.  (The
    // first build divided the tile index by the tile grid for every tile it touched - three times per phase and wave, ~55 scalar instructions at 4 cycles each.)
    const int tiles_x = pin_s(g.tiles_x[pid]), tiles_y = pin_s(g.tiles_y[pid]);
    struct Coord { int n, ty, tx; };
    auto next_of = [&](Coord c) {
        ++c.tx;
        const bool wx = c.tx == tiles_x;
        c.tx = wx ? 0 : c.tx;
        c.ty += wx ? 1 : 0;
        const bool wy = c.ty == tiles_y;
        c.ty = wy ? 0 : c.ty;
        c.n += wy ? 1 : 0;
        return c;
    };
    auto tile_of_coord = [&](const Coord& c) { return px::Tile{c.n, c.ty * TH, c.tx * px::kTW, 0}; };
    Coord cw[4];   // tiles ph - 2, ph - 1, ph, ph + 1
    {
        const px::Tile t0 = px::tile_of(first, 1, tiles_x, tiles_y, TH, BN);
        cw[2] = Coord{pin_s(t0.n), pin_s(t0.y0 / TH), pin_s(t0.x0 / px::kTW)};
        cw[0] = cw[1] = cw[2];
        cw[3] = next_of(cw[2]);
    }
    ResP p;
    p.H = pin_s(pa.H); p.W = pin_s(pa.W); p.Cin = pin_s(pa.Cin); p.ld_in = pin_s(pa.ld_in); p.Cout = pin_s(pa.Cout - n0); p.ld_out = pin_s(pa.ld_out);
    p.ld_res = pin_s(pa.ld_res); p.act1 = pin_s(pa.act1); p.act2 = pin_s(pa.act2);
    p.alpha1 = __builtin_bit_cast(float, pin_s(__builtin_bit_cast(int, pa.alpha1)));   // (pinned like the integers: an un-pinned kernel-argument field is re-read - s_load + wait -
    p.alpha2 = __builtin_bit_cast(float, pin_s(__builtin_bit_cast(int, pa.alpha2)));   //  at every use: 800-950 cycles per fp32 output in the first build)
    constexpr int ES = F32 ? 4 : 2;   // bytes per output / shortcut element
    p.bias = pin_p(pa.bias ? (g_cfloat*)pa.bias + n0 : (g_cfloat*)nullptr);
    p.out = pin_p((g_half*)((__attribute__((address_space(1))) char*)pa.out + (size_t)n0 * ES));
    p.res = pin_p(pa.residual ? (g_chalf*)((__attribute__((address_space(1))) const char*)pa.residual + (size_t)n0 * ES) : (g_chalf*)nullptr);

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pa.in), 0, g.in_bytes[pid], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pa.wgt), 0, g.w_bytes[pid], 0x00020000);
    char* const Bs = smem;
    char* const patches = smem + B_BYTES;   // two buffers: tile t's patch lies in buffer t & 1

    // ---- the weights, once: k-step e = tap * KC + kc starts at k = 32 e of a packed row; piece = 16 rows x 64 bytes, lane-linear, chunks swizzled on the source side
    {
        const int lrow = lane >> 2;
        const int clog = (lane & 3) ^ px::swz32(lrow);   // (the swizzle key of row 16 rb + lrow is lrow's)
        const int kpad = pa.Kpad;
#pragma unroll 1
        for (int piece = wave_all; piece < B_PIECES; piece += NW) {
            const int e = piece / NFRAG, rb = piece - e * NFRAG;
            const unsigned voff = (unsigned)(((n0 + rb * 16 + lrow) * kpad + e * 32 + clog * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + piece * 1024), 16, voff, 0, 0, 0);
        }
    }
    // ---- a tile's patch: the pieces of a plane are dealt to the four waves of the fetching group (piece = wave + 4 k).  What a lane fetches for piece k does not
    // depend on the tile: patch pixel (py, px), logical chunk - computed once; per tile the two border compares and the address remain.
    // Padding ring, pitch padding and ragged channels are range-checked away (zero fill): a lane whose chunk never exists carries py = 255.
    int d_pos[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const px::DmaLane d = px::dma_lane(wave + 4 * k, lane);
        d_pos[k] = (d.px < px::kPW ? d.py : 255) | (d.px << 8) | (d.clog << 16);
    }
    // The patch goes global -> registers -> LDS (buffer_load_dwordx4, later ds_write_b128 into the lane-linear image an LDS-DMA piece would have written): a
    // buffer_load ... lds piece costs its wave 140-280 cycles of issue here (1.1-2.3k cycles for a tile's 8 pieces per wave with two or three waves per SIMD,
    // profiles/r06_res3_anatomy.txt), a register load a few; the round trip passes under the epilogue arithmetic that stands between fetch and commit.
    // Every non-multiplying role fetches its share of the pieces (piece c of the NPW = KC * PPW a wave owns goes to part c mod NPARTS: 16 registers in flight
    // per wave instead of 32 when two roles share the work).
    constexpr int NPW = KC * PPW;   // pieces per wave and tile
    auto piece_voff = [&](const px::Tile& Tt, int origin, int c) {
        const int kc = c / PPW, k = c - kc * PPW;
        const int py = d_pos[k] & 255, pxx = (d_pos[k] >> 8) & 255, clog = d_pos[k] >> 16;
        const bool ok = ((unsigned)(Tt.y0 - 1 + py) < (unsigned)p.H) & ((unsigned)(Tt.x0 - 1 + pxx) < (unsigned)p.W) & (kc * 32 + clog * 8 < p.Cin);
        return ok ? (unsigned)((origin + (py * p.W + pxx) * p.ld_in + clog * 8 + kc * 32) * 2) : kOOB;
    };
    auto piece_lds = [&](char* dst, int c) {   // where piece c lands, or nullptr for a piece beyond the plane (wave-uniform)
        const int kc = c / PPW, k = c - kc * PPW, piece = wave + 4 * k;
        return (PIECES % 4 != 0 && k == PPW - 1 && piece >= PIECES) ? (char*)nullptr : dst + kc * PLANE + piece * 1024 + lane * 16;
    };
    auto patch_origin = [&](const px::Tile& Tt) { return ((Tt.n * p.H + Tt.y0 - 1) * p.W + Tt.x0 - 1) * p.ld_in; };   // patch pixel (0, 0), in halfs
    if (grp_id == 0) {   // tile 0's patch (tile 1's is fetched in phase 0, and so on)
        const px::Tile T0 = tile_of_coord(cw[2]);
        const int origin = patch_origin(T0);
        intx4 pr[NPW];
#pragma unroll
        for (int c = 0; c < NPW; ++c) pr[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, piece_voff(T0, origin, c), 0, 0);
#pragma unroll
        for (int c = 0; c < NPW; ++c)
            if (char* d = piece_lds(patches, c)) *reinterpret_cast<intx4*>(d) = pr[c];
    }

    // this lane's bias values (fp16: channels {0, 8, 4, 12}[g] + [0, 4) of every fragment; fp32: channels 4 g + [0, 4), where the sums start), fragment read offsets
    const int grp = lane >> 4;
    floatx4 bias[NFRAG];
#pragma unroll
    for (int j = 0; j < NFRAG; ++j) {
        bias[j] = F32 ? floatx4{0.f, 0.f, 0.f, 0.f} : floatx4{-0.f, -0.f, -0.f, -0.f};   // fp16: x + (-0.0f) == x, bit for bit, for every x
        if (p.bias) bias[j] = *reinterpret_cast<__attribute__((address_space(1))) const floatx4*>(p.bias + j * 16 + (F32 ? grp * 4 : (((grp & 1) << 3) | ((grp & 2) << 1))));
    }
    int a_off[3];   // tile row wave * MI, tap column q; a further tile row / filter row adds kRowStepBytes (the swizzle key is unchanged: kPitch % 8 == 0)
#pragma unroll
    for (int q = 0; q < 3; ++q) a_off[q] = px::frag_offset(wave * MI, lane & 15, 0, q, grp);
    const int brow = F32 ? (lane & 15) : sigma16(lane & 15);   // (fp32 fragments are 16-byte stores as they are: no row permutation)
    const int fb_off = brow * 64 + ((grp ^ px::swz32(brow)) << 4);   // + j * 1024 + e * BSTEP

    floatx4 acc[MI][NFRAG];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // the shortcut's values of the tile a group is multiplying: requested at the head of its k-loop, used in the next phases
    constexpr int NP = NFRAG / 2;
    half8 rres[F32 ? 1 : MI][NP > 0 ? NP : 1];
    half4 rlast[F32 ? 1 : MI];
    floatx4 rresf[F32 ? MI : 1][F32 ? NFRAG : 1];
#pragma unroll
    for (int i = 0; i < (F32 ? 1 : MI); ++i) {
        rlast[i] = half4{0, 0, 0, 0};
#pragma unroll
        for (int jp = 0; jp < (NP > 0 ? NP : 1); ++jp) rres[i][jp] = half8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < (F32 ? MI : 1); ++i)
#pragma unroll
        for (int j = 0; j < (F32 ? NFRAG : 1); ++j) rresf[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int cpair = 16 * (lane >> 5) + 8 * (grp & 1);
    const int clast = (NFRAG - 1) * 16 + (((grp & 1) << 3) | ((grp & 2) << 1));

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the weights and the first patch have landed

    // the fragments of k-step e: MI pixel fragments of the patch, NFRAG channel fragments of the weight slab (16 bytes per lane: 8 halfs / 4 floats)
    typedef std::conditional_t<F32, floatx4, half8> frag_t;
    auto read_step = [&](const char* patch, int e, frag_t (&af)[MI], frag_t (&bf)[NFRAG]) {
        const int tap = e / KC, kc = e - tap * KC, r = tap / 3, q = tap - 3 * r;
        const char* pa_ = patch + kc * PLANE + r * px::kRowStepBytes + a_off[q];
        const char* pb_ = Bs + e * BSTEP + fb_off;
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) bf[j] = *reinterpret_cast<const frag_t*>(pb_ + j * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const frag_t*>(pa_ + i * px::kRowStepBytes);
    };
    // row slabs [i0, i1) of tile Tp from this wave's accumulators to memory
    auto finish = [&](const px::Tile& Tp, int i0, int i1) {
        const int x = Tp.x0 + (lane & 15);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i < i0 || i >= i1) continue;
            const int y = Tp.y0 + wave * MI + i;
            const bool okpix = (y < p.H) & (x < p.W);
            const int m = okpix ? (Tp.n * p.H + y) * p.W + x : 0;
            if constexpr (F32) {
                res_epilogue_slab_f32<NFRAG>(p, acc[i], rresf[i], okpix, m, lane);
            } else {
                constexpr int ih = F32 ? 0 : 1;   // (index the fp16 shortcut registers only where they exist)
                if (p.act1 == ACT_SILU) res_epilogue_slab<NFRAG, ACT_SILU>(p, acc[i], bias, rres[i * ih], rlast[i * ih], okpix, m, lane);
                else if (p.act1 == ACT_RELU) res_epilogue_slab<NFRAG, ACT_RELU>(p, acc[i], bias, rres[i * ih], rlast[i * ih], okpix, m, lane);
                else res_epilogue_slab<NFRAG, ACT_NONE>(p, acc[i], bias, rres[i * ih], rlast[i * ih], okpix, m, lane);
            }
        }
    };

    // ---- phases.  Group c = ph mod NG multiplies tile ph in phase ph, out of patch buffer ph & 1.  The group that multiplied in phase ph - 1 ("role 1")
    // finishes its tile from its registers; with three groups it finishes the first half of the row slabs now and the second half in the phase after ("role 2").
    // The vector ALU is 16 lanes wide: 4 cycles per instruction and wave, 16 for exp / rcp - 53 cycles per SiLU output and wave, 1.7k cycles of a SIMD's VALU for
    // the 32 outputs per lane of a tile, next to 2.3k cycles of its matrix pipe: the epilogue has to run BESIDE the MFMAs, on other waves, and a lone epilogue wave
    // per SIMD did not keep up (3.0k cycles).  Every finishing role also fetches its share of the patch of tile ph + 1 into the other buffer - free since the
    // barrier that ended phase ph - 1: every fragment read of that k-loop had completed (lgkmcnt(0)).  One s_barrier per phase; the patch is written
    // (lgkmcnt(0) of the writing waves) before the barrier that lets the next phase's multiplying group read it.
    constexpr int SPLIT = NG == 3 ? (MI + 1) / 2 : MI;   // row slabs finished in role 1
    int role = grp_id == 0 ? 0 : NG - grp_id;            // (ph - grp_id) mod NG at ph = 0
    for (int ph = 0; ph < T + NG - 1; ++ph) {
        TRTX_RES_STAMP(ph, 0);
        if (role == 0) {
            if (ph < T && !(TRTX_RES_ABLATE & 4)) {
                const char* patch = patches + (ph & 1) * PATCH_BYTES;
                if (p.res) {   // (unconditional loads from clamped addresses: a conditional load waits where it stands)
                    const px::Tile Tc = tile_of_coord(cw[2]);
                    const int x = Tc.x0 + (lane & 15);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int y = Tc.y0 + wave * MI + i;
                        const bool okpix = (y < p.H) & (x < p.W);
                        const size_t mrow = (size_t)(okpix ? (Tc.n * p.H + y) * p.W + x : 0) * p.ld_res;
                        if constexpr (F32) {
#pragma unroll
                            for (int j = 0; j < NFRAG; ++j) {
                                const int co = 16 * j + 4 * grp;
                                rresf[i][j] = *reinterpret_cast<__attribute__((address_space(1))) const floatx4*>(p.resf() + mrow + (co < p.Cout ? co : 0));
                            }
                        } else {
                            constexpr int ih = F32 ? 0 : 1;
                            g_chalf* rrow = p.res + mrow;
#pragma unroll
                            for (int jp = 0; jp < NP; ++jp) {
                                const int co = 32 * jp + cpair;
                                rres[i * ih][jp] = *reinterpret_cast<__attribute__((address_space(1))) const half8*>(rrow + (co < p.Cout ? co : 0));
                            }
                            if constexpr ((NFRAG & 1) != 0) rlast[i * ih] = *reinterpret_cast<__attribute__((address_space(1))) const half4*>(rrow + (clast < p.Cout ? clast : 0));
                        }
                    }
                }
                if constexpr (F32) {
                    // two-level K sum, as conv_igemm_tile<..., F32>::compute(): the running total starts at the bias, every 16-channel step sums from zero
                    // (srcC = 0 on its first MFMA) and its partial joins the total by a VALU add that stands in front of the MFMA restarting the partial
                    // Two partial-sum register sets: the partial of step e - 1 joins the total AFTER the first MFMAs of step e have been issued - with one or two
                    // fragments per wave (the 16- / 32-wide column tiles) an add placed right behind the partial's last MFMA waits out that MFMA (40 cycles
                    // + the read hazard) with the matrix pipe idle: 0.60 of the fp32 peak instead of the 0.8 the loop can reach.  Same additions, same order.
                    floatx4 part[2][MI][NFRAG];
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j) {
                            acc[i][j] = bias[j];
                            part[1][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};   // ("step -1": adds zero, as conv_igemm_tile's first add does)
                        }
                    floatx4 af[2][MI], bf[2][NFRAG];
                    read_step(patch, 0, af[0], bf[0]);
#pragma unroll
                    for (int e = 0; e < NK; ++e) {
                        if (e + 1 < NK) read_step(patch, e + 1, af[(e + 1) & 1], bf[(e + 1) & 1]);
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                            for (int i = 0; i < MI; ++i)
                                part[e & 1][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[e & 1][j][0], af[e & 1][i][0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler hoists the adds back behind the previous step's last MFMA: `s_nop 8` + its latency)
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                            for (int i = 0; i < MI; ++i) acc[i][j] += part[(e + 1) & 1][i][j];   // the previous step's partial: complete since two MFMA slots ago
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s4 = 1; s4 < 4; ++s4)
#pragma unroll
                            for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                                for (int i = 0; i < MI; ++i)
                                    part[e & 1][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[e & 1][j][s4], af[e & 1][i][s4], part[e & 1][i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);   // (... and it sinks this step's last MFMAs below the next step's first ones, in front of the add that reads them)
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j) acc[i][j] += part[(NK - 1) & 1][i][j];   // the last step's partial
                } else if constexpr (PIPE) {
                    // k-loop.  PIPE: the fragments of step e + 1 are requested BEFORE the MFMAs of step e are issued (an LDS read comes back after ~100 cycles, an MFMA
                    // issues every 16: with one multiplying wave per SIMD nothing else covers a read issued right in front of its use - 4.3k cycles per tile against 3.0k)
                    half8 af[2][MI], bf[2][NFRAG];
                    read_step(patch, 0, af[0], bf[0]);
#pragma unroll
                    for (int e = 0; e < NK; ++e) {
                        if (e + 1 < NK) read_step(patch, e + 1, af[(e + 1) & 1], bf[(e + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                            for (int i = 0; i < MI; ++i)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[e & 1][j], af[e & 1][i], e == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < NK; ++e) {
                        half8 af[MI], bf[NFRAG];
                        read_step(patch, e, af, bf);
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                            for (int i = 0; i < MI; ++i)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], e == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
                    }
                }
            }
            TRTX_RES_STAMP(ph, 1);
        } else {
            // a finishing role: request its share of the next tile's patch, finish its share of the row slabs, write the patch pieces
            auto other = [&](auto PART) {
                constexpr int part = decltype(PART)::value, NPARTS = NG - 1, MINE = (NPW - part + NPARTS - 1) / NPARTS;
                const bool fetch = ph + 1 < T && !(TRTX_RES_ABLATE & 2);
                const px::Tile Tn = tile_of_coord(cw[3]);
                const int origin = patch_origin(Tn);
                char* dst = patches + ((ph + 1) & 1) * PATCH_BYTES;
                intx4 pr[MINE > 0 ? MINE : 1];
                if (fetch) {
#pragma unroll
                    for (int i = 0; i < MINE; ++i) pr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, piece_voff(Tn, origin, part + i * NPARTS), 0, 0);
                }
                TRTX_RES_STAMP(ph, 1);
                constexpr int back = part + 1;   // this group multiplied tile ph - back
                if (ph >= back && ph - back < T) finish(tile_of_coord(cw[2 - back]), part == 0 ? 0 : SPLIT, part == 0 ? SPLIT : MI);
                TRTX_RES_STAMP(ph, 2);
                if (fetch) {
#pragma unroll
                    for (int i = 0; i < MINE; ++i)
                        if (char* d = piece_lds(dst, part + i * NPARTS)) *reinterpret_cast<intx4*>(d) = pr[i];
                }
            };
            if (NG == 2 || role == 1) other(std::integral_constant<int, 0>{});
            else other(std::integral_constant<int, NG == 3 ? 1 : 0>{});
        }
        role = role + 1 == NG ? 0 : role + 1;
        cw[0] = cw[1]; cw[1] = cw[2]; cw[2] = cw[3]; cw[3] = next_of(cw[3]);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        TRTX_RES_STAMP(ph, 3);
        __builtin_amdgcn_s_barrier();
        TRTX_RES_STAMP(ph, 4);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// 1x1 stride-1 convolutions (plain GEMMs over NHWC rows), tactic ConvArgs::t_ws == 8: YOLOv8n's C2f / SPPF / head projections (block.cpp:79-155, model.cpp:188-251).
// These layers are HBM-bound by a wide margin (128 -> 64 at 80 x 80, batch 32: 79 MB against 3.4 GFLOP - 12.5 us of memory time next to 1.3 us of MFMA and
// ~1 us of epilogue VALU at full rate), and what the implicit-GEMM kernel spends on them is structure: a workgroup barrier, LDS-DMA pieces and a refill of the
// weight tile per k-step.  Here: persistent workgroups of 16 INDEPENDENT waves (four per SIMD: the vector ALU needs that many to issue every 2 cycles - a lone
// wave issues one VALU instruction per 8, profiles/r06_valu_rate.txt); the column tile's weights lie in LDS for the workgroup's whole life (K x BN x 2 bytes
// <= 96 KB, conv_res3's layout); a wave owns 16-pixel row fragments: its A operand comes global -> VGPR in MFMA layout (lane = pixel lane & 15, channels
// 8 (lane >> 4) ... + 8: no cross-wave reuse, so no LDS for it), KCH k-steps per chunk, two chunks of registers, the first chunk of the NEXT row fragment
// requested before this one's epilogue; no barrier after the weights have landed.  Same products in the same order as conv_igemm_tile, the epilogue of
// conv_res3: bit-identical to the other tile shapes.
struct ConvRes1Args {
    ConvArgs p;
    unsigned in_bytes, w_bytes;
    int tiles_n, frags, chunk, per;   // column tiles, 16-pixel row fragments, fragments per XCD, fragments per workgroup (a contiguous run)
};

// F32: fp32 operands, conv_res3's conventions (input side in 2-byte units; a k-step = 16 channels = the same 64-byte rows; two-level K sum starting at the bias;
// the fp32 epilogue).  8 waves: an fp32 step is four MFMAs of 32 cycles per 16 bytes a lane loads - two waves per SIMD keep the matrix pipe and the loads busy.
// TAPS2 (round 6, last): the same kernel for the THIN 3x3 layers - 16 input channels, two filter taps per 32-wide k-step (the implicit GEMM's CinK == 16 packing:
// k = tap * 16 + c), stride 1 or 2, output rows a multiple of 16 pixels wide: lane (pixel p, group g) loads the 8 channels 8 (g & 1) ... of the input pixel under
// tap 2 e + (g >> 1) of its output pixel - border taps and tap 9 out of range, i.e. zero.  YOLOv8n's first C2f block at 160 x 160 (16 -> 16, twice) and the
// 16 -> 32 stride-2 layer in front of it: 0.8 M pixels of 32 bytes - HBM-streaming layers on which the implicit GEMM spends a barrier, DMA pieces and a refill per
// k-step for two MFMAs (36 us against 8 of memory time).  Same operand placement as conv_igemm_tile's two-taps-per-step form: the same bits.
template <int NFRAG, int KCH, int NW, bool F32 = false, bool TAPS2 = false>   // NW waves per workgroup: 16 (128 registers each), 8 for the 128-wide column tile (its 32 bias + 32 accumulator + 32 operand registers)
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_res1_f16_kernel(const ConvRes1Args g) {
    constexpr int BN = 16 * NFRAG;
    constexpr int BSTEP = BN * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // nk * BSTEP bytes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane >> 4;
    const ConvArgs& pa = g.p;
    const int nk = pa.Kpad / 32;                 // k-steps
    const int nch = (nk + KCH - 1) / KCH;        // chunks of KCH k-steps (the last one may be short: its missing steps are range-checked to zero and multiply nothing)
    // workgroup -> (xcd, slot) -> column tile tn = slot % tiles_n, the (slot / tiles_n)-th run of `per` row fragments of the XCD's chunk
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tn = slot % g.tiles_n, run = slot / g.tiles_n;
    const int f_end = min(xcd * g.chunk + g.chunk, g.frags);
    const int f_first = xcd * g.chunk + run * g.per;
    const int nf = f_first < f_end ? min(g.per, f_end - f_first) : 0;
    if (nf == 0) return;
    const int n0 = tn * BN;
    ResP p;
    p.H = 0; p.W = 0; p.Cin = pin_s(pa.Cin); p.ld_in = pin_s(pa.ld_in); p.Cout = pin_s(pa.Cout - n0); p.ld_out = pin_s(pa.ld_out);
    p.ld_res = pin_s(pa.ld_res); p.act1 = pin_s(pa.act1); p.act2 = pin_s(pa.act2);
    p.alpha1 = __builtin_bit_cast(float, pin_s(__builtin_bit_cast(int, pa.alpha1))); p.alpha2 = __builtin_bit_cast(float, pin_s(__builtin_bit_cast(int, pa.alpha2)));
    constexpr int ES = F32 ? 4 : 2;   // bytes per output / shortcut element
    p.bias = pin_p(pa.bias ? (g_cfloat*)pa.bias + n0 : (g_cfloat*)nullptr); p.out = pin_p((g_half*)((__attribute__((address_space(1))) char*)pa.out + (size_t)n0 * ES));
    p.res = pin_p(pa.residual ? (g_chalf*)((__attribute__((address_space(1))) const char*)pa.residual + (size_t)n0 * ES) : (g_chalf*)nullptr);
    const int M = pin_s(pa.M);

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pa.in), 0, g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pa.wgt), 0, g.w_bytes, 0x00020000);
    {   // the column tile's weights, once (conv_res3's image: k-step e = 64-byte rows of the BN channels, chunks swizzled on the source side)
        const int lrow = lane >> 2;
        const int clog = (lane & 3) ^ px::swz32(lrow);
        const int kpad = pa.Kpad, pieces = nk * NFRAG;
#pragma unroll 1
        for (int piece = wave; piece < pieces; piece += NW) {
            const int e = piece / NFRAG, rb = piece - e * NFRAG;
            const unsigned voff = (unsigned)(((n0 + rb * 16 + lrow) * kpad + e * 32 + clog * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(smem + piece * 1024), 16, voff, 0, 0, 0);
        }
    }
    floatx4 bias[NFRAG];
#pragma unroll
    for (int j = 0; j < NFRAG; ++j) {
        bias[j] = F32 ? floatx4{0.f, 0.f, 0.f, 0.f} : floatx4{-0.f, -0.f, -0.f, -0.f};
        if (p.bias) bias[j] = *reinterpret_cast<__attribute__((address_space(1))) const floatx4*>(p.bias + j * 16 + (F32 ? grp * 4 : (((grp & 1) << 3) | ((grp & 2) << 1))));
    }
    const int brow = F32 ? (lane & 15) : sigma16(lane & 15);
    const int fb_off = brow * 64 + ((grp ^ px::swz32(brow)) << 4);
    constexpr int NP = NFRAG / 2;
    const int cpair = 16 * (lane >> 5) + 8 * (grp & 1);
    const int clast = (NFRAG - 1) * 16 + (((grp & 1) << 3) | ((grp & 2) << 1));
    const int cmax = p.Cin - grp * 8;   // this lane's 8 channels of k-step e exist while 32 e < cmax
    const int f_stop = f_first + nf;

    // (TAPS2) geometry of the map: fragments per output row, input extent, stride, padding
    const int fpr = TAPS2 ? pin_s(pa.Wo >> 4) : 1, Hin = TAPS2 ? pin_s(pa.H) : 0, Win = TAPS2 ? pin_s(pa.W) : 0, Hout = TAPS2 ? pin_s(pa.Ho) : 1;
    const int cstride = TAPS2 ? pin_s(pa.stride_h) : 1, cpad = TAPS2 ? pin_s(pa.pad_h) : 0;
    // chunk c of row fragment f: KCH 16-byte loads per lane (pixel 16 f + (lane & 15), channels 32 e + 8 (lane >> 4) ...)
    auto load_chunk = [&](int f, int c, intx4 (&a)[KCH]) {
        if constexpr (TAPS2) {
            const int row = f / fpr, x0 = (f - row * fpr) << 4;   // (scalar: a fragment lies inside one output row)
            const int n = row / Hout, yo = row - n * Hout;
            const int xi0 = (x0 + (lane & 15)) * cstride - cpad, yi0 = yo * cstride - cpad;
            const bool okf = f < f_stop;
#pragma unroll
            for (int s = 0; s < KCH; ++s) {
                const int e = c * KCH + s;
                const int tap = 2 * e + (grp >> 1);            // 0 .. 9 (9: the padding of K = 144 to 160)
                const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;   // tap / 3, tap % 3 for tap <= 10
                const int yi = yi0 + dy, xi = xi0 + dx;
                const bool ok = okf & (tap < 9) & (e < nk) & ((unsigned)yi < (unsigned)Hin) & ((unsigned)xi < (unsigned)Win);
                const unsigned voff = ok ? (unsigned)((((n * Hin + yi) * Win + xi) * p.ld_in + (grp & 1) * 8) * 2) : kOOB;
                a[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0);
            }
            return;
        }
        const int m = f * 16 + (lane & 15);
        const unsigned base = ((m < M) & (f < f_stop)) ? (unsigned)((m * p.ld_in + grp * 8) * 2) : kOOB;   // (kOOB + any channel offset stays out of range)
#pragma unroll
        for (int s = 0; s < KCH; ++s) {
            const int e = c * KCH + s;
            const unsigned voff = (32 * e < cmax && e < nk) ? base + (unsigned)(e * 64) : kOOB;
            a[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0);
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the weights have landed; from here on the waves are on their own

    // The wave's work is a flat sequence of chunks (fragment f, chunk c), fragments f_first + wave, + NW, ...: two register sets, the chunk after the one being
    // multiplied always in flight - across fragment boundaries too, so a fragment's epilogue runs with the next fragment's first chunk on its way.  (Round 6,
    // second form: written as a loop over fragments with an inner loop over chunk pairs the compiler carried the register sets through copies behind
    // `s_waitcnt vmcnt(0)` and reused in-flight destination registers as address temporaries - one chunk in flight per wave at best.)
    floatx4 acc[NFRAG];
    floatx4 part[F32 ? 2 : 1][NFRAG];   // fp32: the k-steps' partial sums (conv_res3's two-level sum with the delayed add)
    half8 rv[NP > 0 ? NP : 1];
    half4 rl = half4{0, 0, 0, 0};
    floatx4 rvf[F32 ? NFRAG : 1];
    bool okpix = false;
    int mm = 0;
#pragma unroll
    for (int jp = 0; jp < (NP > 0 ? NP : 1); ++jp) rv[jp] = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < (F32 ? NFRAG : 1); ++j) rvf[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // first chunk of a fragment: fresh sums, the shortcut's values for this fragment's stores (unconditional loads from clamped addresses)
    auto begin = [&](int f) {
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) {
            acc[j] = F32 ? bias[j] : floatx4{0.f, 0.f, 0.f, 0.f};
            part[0][j] = floatx4{0.f, 0.f, 0.f, 0.f};
            part[F32 ? 1 : 0][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
        const int m = f * 16 + (lane & 15);
        okpix = m < M;
        mm = okpix ? m : 0;
        if constexpr (F32) {
            if (p.res) {
                g_cfloat* rrow = p.resf() + (size_t)mm * p.ld_res;
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) {
                    const int co = 16 * j + 4 * grp;
                    rvf[j] = *reinterpret_cast<__attribute__((address_space(1))) const floatx4*>(rrow + (co < p.Cout ? co : 0));
                }
            }
        } else if (p.res) {
            g_chalf* rrow = p.res + (size_t)mm * p.ld_res;
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                const int co = 32 * jp + cpair;
                rv[jp] = *reinterpret_cast<__attribute__((address_space(1))) const half8*>(rrow + (co < p.Cout ? co : 0));
            }
            if constexpr ((NFRAG & 1) != 0) rl = *reinterpret_cast<__attribute__((address_space(1))) const half4*>(rrow + (clast < p.Cout ? clast : 0));
        }
    };
    auto multiply = [&](const intx4 (&a)[KCH], int c) {
#pragma unroll
        for (int s = 0; s < KCH; ++s) {
            const int e = c * KCH + s;
            if constexpr (F32) {
                if (e < nk) {   // (e & 1 == s & 1: KCH is even)
                    const floatx4 af = __builtin_bit_cast(floatx4, a[s]);
                    floatx4 bf[NFRAG];
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) bf[j] = *reinterpret_cast<const floatx4*>(smem + e * BSTEP + j * 1024 + fb_off);
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) part[s & 1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][0], af[0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) acc[j] += part[(s + 1) & 1][j];   // the previous step's partial (zero in front of the first step)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s4 = 1; s4 < 4; ++s4)
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j) part[s & 1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s4], af[s4], part[s & 1][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (e < nk) {
                const half8 af = __builtin_bit_cast(half8, a[s]);
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) {
                    const half8 bf = *reinterpret_cast<const half8*>(smem + e * BSTEP + j * 1024 + fb_off);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, af, acc[j], 0, 0, 0);
                }
            }
        }
    };
    auto finish = [&]() {
        if constexpr (F32) {
            static_assert(KCH % 2 == 0, "the partial-sum register set of a step is chosen by its index in the chunk");
            // the last step's partial.  (Two branches, kept apart by the empty asm: written as a select the compiler turns the register arrays into a
            // dynamically indexed scratch buffer.)
            if (nk & 1) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) acc[j] += part[0][j];
            } else {
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) acc[j] += part[F32 ? 1 : 0][j];
            }
            res_epilogue_slab_f32<NFRAG>(p, acc, rvf, okpix, mm, lane);
        } else {
            if (p.act1 == ACT_SILU) res_epilogue_slab<NFRAG, ACT_SILU>(p, acc, bias, rv, rl, okpix, mm, lane);
            else if (p.act1 == ACT_RELU) res_epilogue_slab<NFRAG, ACT_RELU>(p, acc, bias, rv, rl, okpix, mm, lane);
            else res_epilogue_slab<NFRAG, ACT_NONE>(p, acc, bias, rv, rl, okpix, mm, lane);
        }
    };
    // one chunk: (f, c) with the operands in `a`
    auto process = [&](int f, int c, const intx4 (&a)[KCH]) {
        if (c == 0) begin(f);
        multiply(a, c);
        if (c == nch - 1) finish();
    };
    intx4 a0[KCH], a1[KCH];
    int f = f_first + wave, c = 0;   // the chunk in a0
    if (f >= f_stop) return;
    load_chunk(f, 0, a0);
    for (;;) {
        // the chunk after (f, c) -> a1
        int f1 = f, c1 = c + 1;
        if (c1 == nch) { f1 = f + NW; c1 = 0; }
        // (requested UNCONDITIONALLY - beyond the wave's last fragment every address is out of range and nothing moves: a load behind a branch makes the
        // compiler's wait counts assume the path without it, and every wait for the older chunk then waits for the younger one too)
        load_chunk(f1, c1, a1);
        process(f, c, a0);
        if (f1 >= f_stop) break;
        // ... and the one after that -> a0, whose MFMAs have been issued
        int f2 = f1, c2 = c1 + 1;
        if (c2 == nch) { f2 = f1 + NW; c2 = 0; }
        load_chunk(f2, c2, a0);
        process(f1, c1, a1);
        if (f2 >= f_stop) break;
        f = f2; c = c2;
    }
}

template <int NFRAG, int NW, bool F32 = false, bool TAPS2 = false>
int32_t launch_res1(const ConvRes1Args& g, int lds_bytes, hipStream_t s) {
    // more than 64 KB of dynamic LDS has to be asked for once per device
    static bool asked[64] = {};
    int dev = 0;
    TRTX_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !asked[dev]) {
        TRTX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_res1_f16_kernel<NFRAG, 4, NW, F32, TAPS2>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        asked[dev] = true;
    }
    const int runs = (g.chunk + g.per - 1) / g.per;
    TRTX_LAUNCH((conv_res1_f16_kernel<NFRAG, 4, NW, F32, TAPS2>), dim3(runs * g.tiles_n * 8), dim3(NW * 64), lds_bytes, s, g);
    return TRTX_OK;
}

template <int NFRAG, int KC, int MI, int NG, bool PIPE, int WPS, bool F32 = false>
void launch_res3(const ConvResArgs& g, hipStream_t s) {
    TRTX_LAUNCH((conv_res3_f16_kernel<NFRAG, KC, MI, NG, PIPE, WPS, F32>), dim3(g.slot_start[g.n] * 8), dim3(NG * 256), 0, s, g);
}

struct ResShape {
    int nfrag, kc, mi, wpc;   // column fragments, 32-channel slices, row fragments per wave, workgroups per CU
};
// the instantiation that serves a layer, or nfrag == 0
ResShape res_shape(const ConvArgs& a) {
    ResShape r{0, 0, 0, 0};
    if (a.f32) {
        // fp32: planes of 16 channels; the column tile may be narrower than Cout (a workgroup is bound to one column tile): the widest patch (8 output rows,
        // then 4) that fits beside the tile's weights.  Instantiated: 16 -> 16 (one plane), 32 -> 32, 64 -> 64 as 2 x 32 or 4 x 16 columns, 64 -> 80 and
        // 80 -> 80 as 5 x 16 columns
        if (a.CinK % 16 || a.bn % 16 || a.bn <= 0 || a.Cout_pad % a.bn) return r;
        const int kc = a.CinK / 16, nf = a.bn / 16;
        for (int mi = 2; mi >= 1; --mi) {
            const long lds = 9L * kc * a.bn * 64 + 2L * kc * (4 * mi + 2) * px::kPitch * px::kPixelBytes;
            if (lds > 163840) continue;
            const bool have = (nf == 1 && kc == 1 && mi == 2) || (nf == 2 && kc == 2 && mi == 2) || (nf == 2 && kc == 4 && mi == 1) || (nf == 1 && kc == 4 && mi == 2) ||
                              (nf == 1 && kc == 5 && mi == 1) || (nf == 1 && kc == 2 && mi == 2);
            if (have) return ResShape{nf, kc, mi, 1};
        }
        return r;
    }
    if (a.CinK % 32 || a.bn % 16 || a.bn != a.Cout_pad) return r;
    const int kc = a.CinK / 32, nf = a.bn / 16;
    if (kc == 1 && nf == 2) r = a.H >= 16 ? ResShape{2, 1, 4, 1} : ResShape{2, 1, 2, 2};
    else if (kc == 2 && nf == 4) r = ResShape{4, 2, 2, 1};
    else if (kc == 2 && nf == 5) r = ResShape{5, 2, 2, 1};
    else if (kc == 1 && nf == 4) r = ResShape{4, 1, 2, 1};
    else if (kc == 2 && nf == 2) r = ResShape{2, 2, 2, 1};
    return r;
}

}  // namespace

bool conv_res_possible(const ConvArgs& a) {
    if (a.up_C || a.in_i8 || a.out_i8 || a.res_i8 || a.scalar_out) return false;
    if (a.kh != 3 || a.kw != 3 || a.stride_h != 1 || a.stride_w != 1 || a.pad_h != 1 || a.pad_w != 1 || a.dil_h != 1 || a.dil_w != 1 || a.groups != 1) return false;
    if (a.f32) {   // fp32 engines (conv_igemm_f32.hip's layouts: 16-channel k-steps, 4-byte elements; every activation the fp32 epilogue knows)
        if (a.bk != 16 || a.CinK % 16 || a.Cin % 4 || a.Cin > a.CinK || a.Kpad != 9 * a.CinK || a.Ho != a.H || a.Wo != a.W) return false;
        if (a.Cout % 4 || a.ld_out % 4 || a.ld_in % 4 || (a.residual && a.ld_res % 4)) return false;
        if ((double)a.N * a.H * a.W * a.ld_in * 4.0 >= 2.0e9 || (double)a.N * a.H * a.W * a.ld_out >= 2.0e9) return false;
        return res_shape(a).nfrag != 0;
    }
    if (a.bk != 32 || a.CinK % 32 || a.Cin % 8 || a.Cin > a.CinK || a.Kpad != 9 * a.CinK || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.Cout % 8 || a.ld_out % 8 || a.ld_in % 8 || (a.residual && a.ld_res % 8)) return false;
    if (!(a.bm == 0 || a.bm == 128) || a.t_r3 != 0) return false;
    if (!(a.act1 == ACT_NONE || a.act1 == ACT_RELU || a.act1 == ACT_SILU) || !(a.act2 == ACT_NONE || a.act2 == ACT_RELU)) return false;
    if ((double)a.N * a.H * a.W * a.ld_in * 2.0 >= 2.0e9 || (double)a.N * a.H * a.W * a.ld_out * 2.0 >= 4.0e9) return false;
    return res_shape(a).nfrag != 0;
}

bool conv_res_group_possible(const ConvArgs* a, int n) {
    if (n < 1 || n > kResMax) return false;
    const ResShape r0 = res_shape(a[0]);
    for (int k = 0; k < n; ++k) {
        if (!conv_res_possible(a[k])) return false;
        const ResShape r = res_shape(a[k]);
        if (r.nfrag != r0.nfrag || r.kc != r0.kc || r.mi != r0.mi || a[k].f32 != a[0].f32) return false;
    }
    return true;
}

int32_t conv_res_f16(const ConvArgs* a, int n, hipStream_t s) {
    if (!conv_res_group_possible(a, n)) return TRTX_ERR_UNSUPPORTED;
    const ResShape r = res_shape(a[0]);
    const int th = 4 * r.mi;
    ConvResArgs g{};
    g.n = n;
    int order[kResMax];   // the big maps first
    for (int k = 0; k < n; ++k) order[k] = k;
    std::sort(order, order + n, [&](int x, int y) {
        const long mx = (long)a[x].N * a[x].H * a[x].W, my = (long)a[y].N * a[y].H * a[y].W;
        return mx != my ? mx > my : x < y;
    });
    long sum_chunk = 0;
    for (int k = 0; k < n; ++k) {
        const ConvArgs& ak = a[order[k]];
        g.p[k] = ak;
        g.p[k].M = ak.N * ak.Ho * ak.Wo;
        g.tiles_x[k] = (ak.W + 15) / 16;
        g.tiles_y[k] = (ak.H + th - 1) / th;
        g.tiles[k] = ak.N * g.tiles_y[k] * g.tiles_x[k];
        g.chunk[k] = (g.tiles[k] + 7) / 8;
        g.tiles_n[k] = ak.Cout_pad / ak.bn;
        const int es = ak.f32 ? 4 : 2;
        g.in_bytes[k] = (unsigned)((((size_t)ak.N * ak.H * ak.W - 1) * ak.ld_in + ak.Cin) * es);
        g.w_bytes[k] = (unsigned)((size_t)ak.Cout_pad * ak.Kpad * es);
        if (ak.f32) {   // the kernel's view of an fp32 launch: the input side in 2-byte units (conv_igemm_f32.hip kernel_units)
            g.p[k].Cin = 2 * ak.Cin; g.p[k].ld_in = 2 * ak.ld_in; g.p[k].CinK = 2 * ak.CinK; g.p[k].K = 2 * ak.K; g.p[k].Kpad = 2 * ak.Kpad;
        }
        sum_chunk += (long)g.chunk[k] * g.tiles_n[k];
    }
    // workgroup slots per XCD (32 CUs x workgroups per CU), shared out in proportion to the problems' tiles; every workgroup of a problem then takes a
    // contiguous run of `per` tiles of its XCD's chunk
    const int cap = 32 * r.wpc;
    int slots = 0;
    for (int k = 0; k < n; ++k) {
        // (column tiles: the problem's slots are dealt round-robin to its tiles_n column tiles, each of which walks all of the XCD's row tiles)
        int share = (int)std::max<long>(1, (long)cap * g.chunk[k] * g.tiles_n[k] / std::max<long>(1, sum_chunk) / g.tiles_n[k]);
        share = std::min(share, g.chunk[k]);
        g.per[k] = (g.chunk[k] + share - 1) / share;
        const int ns = (g.chunk[k] + g.per[k] - 1) / g.per[k];
        g.slot_start[k] = slots;
        slots += ns * g.tiles_n[k];
    }
    for (int k = n; k <= kResMax; ++k) g.slot_start[k] = slots;
    // <column fragments, channel slices, row fragments per wave, wave groups, pipelined fragment reads, waves per SIMD, fp32>
    if (a[0].f32) {   // two wave groups (8 waves): an fp32 k-loop is 16x an fp16 one, the finishing role has time to spare
        if (r.nfrag == 1 && r.kc == 1) launch_res3<1, 1, 2, 2, true, 2, true>(g, s);
        else if (r.nfrag == 1 && r.kc == 2) launch_res3<1, 2, 2, 2, true, 2, true>(g, s);
        else if (r.nfrag == 2 && r.kc == 2) launch_res3<2, 2, 2, 2, true, 2, true>(g, s);
        else if (r.nfrag == 2 && r.kc == 4) launch_res3<2, 4, 1, 2, true, 2, true>(g, s);
        else if (r.nfrag == 1 && r.kc == 4) launch_res3<1, 4, 2, 2, true, 2, true>(g, s);
        else if (r.nfrag == 1 && r.kc == 5) launch_res3<1, 5, 1, 2, true, 2, true>(g, s);
        else return TRTX_ERR_UNSUPPORTED;
        return check_launch("conv_res3_f32");
    }
    if (r.nfrag == 2 && r.kc == 1 && r.mi == 4) launch_res3<2, 1, 4, 3, true, 3>(g, s);        // 12 waves, one workgroup per CU (72 KB of LDS)
    else if (r.nfrag == 2 && r.kc == 1) launch_res3<2, 1, 2, 2, false, 4>(g, s);               // 8 waves, two workgroups per CU (48 KB each)
    else if (r.nfrag == 4 && r.kc == 2) launch_res3<4, 2, 2, 3, true, 3>(g, s);                // 12 waves (132 KB)
    else if (r.nfrag == 5 && r.kc == 2) launch_res3<5, 2, 2, 2, true, 2>(g, s);                // 8 waves: 208 registers (150 KB)
    else if (r.nfrag == 4 && r.kc == 1) launch_res3<4, 1, 2, 3, true, 3>(g, s);                // 12 waves (66 KB)
    else if (r.nfrag == 2 && r.kc == 2) launch_res3<2, 2, 2, 3, true, 3>(g, s);                // 12 waves (96 KB)
    else return TRTX_ERR_UNSUPPORTED;
    return check_launch("conv_res3_f16");
}


// the thin 3x3 form (TAPS2): 16 input channels packed two taps per k-step, stride 1 / 2, output rows of whole 16-pixel fragments
static bool res1_taps2(const ConvArgs& a) {
    return !a.f32 && a.kh == 3 && a.kw == 3 && a.pad_h == 1 && a.pad_w == 1 && a.dil_h == 1 && a.dil_w == 1 && a.groups == 1 && a.stride_h == a.stride_w &&
           (a.stride_h == 1 || a.stride_h == 2) && a.Cin == 16 && a.CinK == 16 && a.bk == 32 && a.Kpad == 160 && a.Wo % 16 == 0 && a.Wo > 0 &&
           a.Ho == (a.H + 2 - 3) / a.stride_h + 1 && a.Wo == (a.W + 2 - 3) / a.stride_w + 1;
}

bool conv_res1_possible(const ConvArgs& a) {
    if (a.up_C || a.in_i8 || a.out_i8 || a.res_i8 || a.scalar_out) return false;
    if (res1_taps2(a)) {
        if (a.Cout % 8 || a.ld_out % 8 || a.ld_in % 8 || (a.residual && a.ld_res % 8)) return false;
        if (!(a.bm == 0 || a.bm == 128) || a.t_r3 != 0) return false;
        if (!(a.act1 == ACT_NONE || a.act1 == ACT_RELU || a.act1 == ACT_SILU) || !(a.act2 == ACT_NONE || a.act2 == ACT_RELU)) return false;
        if (!(a.bn == 16 || a.bn == 32) || a.Cout_pad % a.bn) return false;
        const double pin = (double)a.N * a.H * a.W, pout = (double)a.N * a.Ho * a.Wo;
        return pin * a.ld_in * 2.0 < 2.0e9 && pout * a.ld_out < 2.0e9 && pout * (a.residual ? a.ld_res : 1) < 2.0e9;
    }
    if (a.kh != 1 || a.kw != 1 || a.stride_h != 1 || a.stride_w != 1 || a.pad_h != 0 || a.pad_w != 0 || a.groups != 1) return false;
    if (a.f32) {   // fp32 engines: 16-channel k-steps, 4-byte elements, every activation the fp32 epilogue knows
        if (a.bk != 16 || a.CinK % 16 || a.Cin % 4 || a.Cin > a.CinK || a.Kpad != a.CinK || a.Ho != a.H || a.Wo != a.W) return false;
        if (a.Cout % 4 || a.ld_out % 4 || a.ld_in % 4 || (a.residual && a.ld_res % 4)) return false;
        if (!(a.bn == 16 || a.bn == 32 || a.bn == 64 || a.bn == 80) || a.Cout_pad % a.bn) return false;
        if ((long)a.Kpad * a.bn * 4 > 98304) return false;
        const double px = (double)a.N * a.H * a.W;
        return px * a.ld_in * 4.0 < 2.0e9 && px * a.ld_out < 2.0e9 && px * (a.residual ? a.ld_res : 1) < 2.0e9;
    }
    if (a.bk != 32 || a.CinK % 32 || a.Cin % 8 || a.Cin > a.CinK || a.Kpad != a.CinK || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.Cout % 8 || a.ld_out % 8 || a.ld_in % 8 || (a.residual && a.ld_res % 8)) return false;
    if (!(a.bm == 0 || a.bm == 128) || a.t_r3 != 0) return false;
    if (!(a.act1 == ACT_NONE || a.act1 == ACT_RELU || a.act1 == ACT_SILU) || !(a.act2 == ACT_NONE || a.act2 == ACT_RELU)) return false;
    if (!(a.bn == 32 || a.bn == 64 || a.bn == 80 || a.bn == 128) || a.Cout_pad % a.bn) return false;
    if ((long)a.Kpad * a.bn * 2 > 98304) return false;   // the column tile's weights in LDS
    const double px = (double)a.N * a.H * a.W;
    return px * a.ld_in * 2.0 < 2.0e9 && px * a.ld_out < 2.0e9 && px * (a.residual ? a.ld_res : 1) < 2.0e9;
}

int32_t conv_res1_f16(const ConvArgs& a, hipStream_t s) {
    if (!conv_res1_possible(a)) return TRTX_ERR_UNSUPPORTED;
    ConvRes1Args g{};
    g.p = a;
    g.p.M = a.N * a.Ho * a.Wo;
    const int es = a.f32 ? 4 : 2;
    g.in_bytes = (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ld_in + a.Cin) * es);
    g.w_bytes = (unsigned)((size_t)a.Cout_pad * a.Kpad * es);
    if (a.f32) {   // the kernel's view: the input side in 2-byte units
        g.p.Cin = 2 * a.Cin; g.p.ld_in = 2 * a.ld_in; g.p.CinK = 2 * a.CinK; g.p.K = 2 * a.K; g.p.Kpad = 2 * a.Kpad;
    }
    g.tiles_n = a.Cout_pad / a.bn;
    g.frags = (g.p.M + 15) / 16;
    g.chunk = (g.frags + 7) / 8;
    // Launch geometry: one persistent workgroup per CU takes an equal, contiguous share of its XCD's row fragments.  (Measured on YOLOv8n b32 against
    // 4- / 8- / 16-wave workgroups of 4-16 fragments per wave - a grid the hardware balances: every variant within the run-to-run spread of three contexts in
    // flight, the persistent form the fastest single context: profiles/r06_engine_ab.txt, block v1.)
    const int nw = a.bn == 128 ? 8 : 16;
    const int runs = std::max(1, 32 / g.tiles_n);   // 32 CUs per XCD, shared by the column tiles
    g.per = std::max(1, (g.chunk + runs - 1) / runs);
    const int lds = (g.p.Kpad / 32) * a.bn * 64;
    int32_t st = TRTX_ERR_UNSUPPORTED;
    (void)nw;
    if (a.f32) {
        switch (a.bn) {
            case 16: st = launch_res1<1, 8, true>(g, lds, s); break;
            case 32: st = launch_res1<2, 8, true>(g, lds, s); break;
            case 64: st = launch_res1<4, 8, true>(g, lds, s); break;
            case 80: st = launch_res1<5, 8, true>(g, lds, s); break;
        }
        return st != TRTX_OK ? st : check_launch("conv_res1_f32");
    }
    if (res1_taps2(a)) {
        if (a.bn == 16) st = launch_res1<1, 16, false, true>(g, lds, s);
        else st = launch_res1<2, 16, false, true>(g, lds, s);
        return st != TRTX_OK ? st : check_launch("conv_res1_taps2_f16");
    }
    switch (a.bn) {
        case 32: st = launch_res1<2, 16>(g, lds, s); break;
        case 64: st = launch_res1<4, 16>(g, lds, s); break;
        case 80: st = launch_res1<5, 12>(g, lds, s); break;   // (12 waves: 170 registers each - at 128 the flat chunk pipeline spills)
        case 128: st = launch_res1<8, 8>(g, lds, s); break;
    }
    return st != TRTX_OK ? st : check_launch("conv_res1_f16");
}

}  // namespace trtx
