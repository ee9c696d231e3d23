// Index arithmetic of the resident-patch 3x3 kernel (conv_igemm.hip, tactic ConvArgs::t_ws == 3): which patch pixel and which
// 8-channel chunk a DMA lane fetches, where it lands in LDS, where a fragment lane reads its 16 bytes, how the weight tile is laid out.
// Plain integer functions, usable from host code: tests/test_patch_index_cpu.py compiles this header with g++ and replays the kernel's data
// path lane by lane (DMA pieces -> LDS image -> ds_read_b128 fragments -> v_mfma_f32_16x16x32_f16 semantics) against a direct convolution, so
// that the part of the kernel a CPU can check is checked before the kernel sees a GPU.  The kernel calls exactly these functions.
//
// Geometry.  An output tile is TH x 16 pixels of one image (TH = 4 * MI: wave w owns tile rows w * MI .. w * MI + MI - 1, one 16-pixel MFMA
// fragment each).  The input patch is (TH + 2) x 18 pixels, stored with a row pitch of 24 pixels; one LDS plane per 32-channel slice, 64 bytes
// per pixel, the four 16-byte chunks of a pixel XOR-swizzled by (pixel >> 1) & 3 (conv_ws.hip's layout: conflict-free for 16 consecutive
// pixels and for row pitches that are multiples of 8 pixels).  A DMA piece (one buffer_load_dwordx4 ... lds of a wave) is 16 consecutive patch
// pixels x 64 bytes = 1 KiB, lane-linear: lane l writes bytes [16 l, 16 l + 16) of the piece.
#pragma once

#if defined(__HIPCC__)
#define TRTX_HD __host__ __device__ __forceinline__
#else
#define TRTX_HD inline
#endif

namespace trtx {
namespace patchidx {

constexpr int kTW = 16;        // tile width in pixels (one MFMA fragment per tile row)
constexpr int kPW = 18;        // patch width: tile + one halo column on each side
constexpr int kPitch = 24;     // LDS row pitch of the patch in pixels (a multiple of 8: a row step leaves the swizzle key unchanged)
constexpr int kPixelBytes = 64;  // one pixel of one 32-channel plane

TRTX_HD int patch_rows(int th) { return th + 2; }
TRTX_HD int plane_pixels(int th) { return patch_rows(th) * kPitch; }       // a multiple of 16 for th = 8, 16
TRTX_HD int plane_bytes(int th) { return plane_pixels(th) * kPixelBytes; }
TRTX_HD int plane_pieces(int th) { return plane_pixels(th) / 16; }

TRTX_HD int swz_key(int pp) { return (pp >> 1) & 3; }

// DMA side: lane `lane` of piece `piece` of a plane fetches LOGICAL chunk `clog` (channels 8 clog .. 8 clog + 7 of the plane's slice) of patch
// pixel (py, px) and writes it at byte piece * 1024 + lane * 16 of the plane, i.e. into PHYSICAL chunk lane & 3 of that pixel.
struct DmaLane {
    int py, px, clog;
};
TRTX_HD DmaLane dma_lane(int piece, int lane) {
    const int pp = piece * 16 + (lane >> 2);
    DmaLane d;
    d.py = pp / kPitch;
    d.px = pp - d.py * kPitch;
    d.clog = (lane & 3) ^ swz_key(pp);
    return d;
}
TRTX_HD int dma_lds_offset(int piece, int lane) { return piece * 1024 + lane * 16; }

// Fragment side: the lane that supplies pixel (oy, ox) of the tile and k-chunk `kchunk` (= lane >> 4: channels 8 kchunk .. + 7 of the slice) to
// the MFMA of tap (r, q) reads 16 bytes at this offset of the plane.
TRTX_HD int frag_offset(int oy, int ox, int r, int q, int kchunk) {
    const int pp = (oy + r) * kPitch + ox + q;
    return pp * kPixelBytes + ((kchunk ^ swz_key(pp)) << 4);
}
// a filter-row step adds this many bytes and leaves the swizzle key as it is (kPitch % 8 == 0)
constexpr int kRowStepBytes = kPitch * kPixelBytes;

// Weight tile of one k-step (32 k values = 64 bytes per output channel): conv_igemm's B-tile layout for 32-wide steps.  Lane `lane` of the pass-j
// piece of wave `wave` (4 waves) fetches logical chunk `clog` of row `row`; lane-linear destination; a fragment lane reads row
// 16 jf + (lane & 15), physical chunk (lane >> 4) ^ swz32(row).
TRTX_HD int swz32(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
struct WLane {
    int row, clog;
};
TRTX_HD WLane w_lane(int pass, int wave, int lane) {
    WLane w;
    w.row = (4 * pass + wave) * 16 + (lane >> 2);
    w.clog = (lane & 3) ^ swz32(lane >> 2);
    return w;
}
TRTX_HD int w_lds_offset(int pass, int wave, int lane) { return (4 * pass + wave) * 1024 + lane * 16; }
TRTX_HD int w_frag_offset(int jf, int lane) {
    const int frow = lane & 15;
    return (jf * 16 + frow) * 64 + (((lane >> 4) ^ swz32(frow)) << 4);
}

// Tiles: column tiles of the same pixels are neighbours (they share the patch in L2), then tiles along x, y, images.
struct Tile {
    int n, y0, x0, n0;
};
TRTX_HD Tile tile_of(int tile, int tiles_n, int tiles_x, int tiles_y, int th, int bn) {
    Tile t;
    const int tn = tile % tiles_n;
    int rest = tile / tiles_n;
    const int tx = rest % tiles_x;
    rest /= tiles_x;
    const int ty = rest % tiles_y;
    t.n = rest / tiles_y;
    t.y0 = ty * th;
    t.x0 = tx * kTW;
    t.n0 = tn * bn;
    return t;
}

}  // namespace patchidx
}  // namespace trtx
