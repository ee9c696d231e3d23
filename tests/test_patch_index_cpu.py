"""The resident-patch 3x3 kernel's index arithmetic (tensorrtx_amd/csrc/kernels/patch_index.h), replayed on the CPU lane by lane: DMA pieces -> LDS image
of the patch planes and the weight tiles -> ds_read_b128 fragments -> v_mfma_f32_16x16x32_f16 semantics, against a direct convolution.  The header is
compiled with g++ as it stands (the kernel calls the same functions); what this file mirrors by hand are the few lines of the kernel that turn a
DMA lane's patch pixel into a global address (bounds -> zero fill) and the k-step order.  Written before the kernel had seen a GPU."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "tensorrtx_amd", "csrc", "kernels")

SHIM = r"""
#include "patch_index.h"
using namespace trtx::patchidx;
extern "C" {
int c_plane_bytes(int th) { return plane_bytes(th); }
int c_plane_pieces(int th) { return plane_pieces(th); }
void c_dma_lane(int piece, int lane, int* out) { DmaLane d = dma_lane(piece, lane); out[0] = d.py; out[1] = d.px; out[2] = d.clog; out[3] = dma_lds_offset(piece, lane); }
int c_frag_offset(int oy, int ox, int r, int q, int kchunk) { return frag_offset(oy, ox, r, q, kchunk); }
int c_row_step() { return kRowStepBytes; }
void c_w_lane(int pass, int wave, int lane, int* out) { WLane w = w_lane(pass, wave, lane); out[0] = w.row; out[1] = w.clog; out[2] = w_lds_offset(pass, wave, lane); }
int c_w_frag_offset(int jf, int lane) { return w_frag_offset(jf, lane); }
void c_tile_of(int tile, int tiles_n, int tiles_x, int tiles_y, int th, int bn, int* out) { Tile t = tile_of(tile, tiles_n, tiles_x, tiles_y, th, bn); out[0] = t.n; out[1] = t.y0; out[2] = t.x0; out[3] = t.n0; }
}
"""


@pytest.fixture(scope="module")
def shim():
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "shim.cpp")
        open(src, "w").write(SHIM)
        so = os.path.join(tmp, "shim.so")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", f"-I{HDR}", src, "-o", so])
        yield ctypes.CDLL(so)


def _conv_ref(x, w):
    """x [N,H,W,C] fp16 values, w [Co,C,3,3] fp16 values: 3x3 stride 1 pad 1 in float64"""
    N, H, W, C = x.shape
    xp = np.zeros((N, H + 2, W + 2, C))
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((N, H, W, w.shape[0]))
    for r in range(3):
        for q in range(3):
            out += np.einsum("nhwc,oc->nhwo", xp[:, r:r + H, q:q + W], w[:, :, r, q].astype(np.float64))
    return out


@pytest.mark.parametrize("N,H,W,Cin,Cout,bn,mi", [(2, 19, 21, 64, 64, 64, 2), (1, 20, 16, 64, 128, 64, 4), (1, 9, 33, 80, 80, 80, 2), (1, 17, 17, 32, 128, 128, 4), (1, 10, 18, 256, 64, 64, 2)])
def test_lane_level_replay_of_the_patch_kernel_is_the_convolution(shim, N, H, W, Cin, Cout, bn, mi):
    L = shim
    rng = np.random.default_rng(N * 1000 + H * 10 + Cin)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float16)
    cink = (Cin + 31) // 32 * 32
    KC, NK, TH = cink // 32, 9 * (cink // 32), 4 * mi
    cout_pad = (Cout + bn - 1) // bn * bn
    # packed weights [Cout_pad][Kpad], k = tap * CinK + c (kernels.h, ConvArgs::wgt)
    packed = np.zeros((cout_pad, 9 * cink), np.float16)
    for r in range(3):
        for q in range(3):
            packed[:Cout, (r * 3 + q) * cink:(r * 3 + q) * cink + Cin] = w[:, :, r, q]
    plane_bytes, pieces = L.c_plane_bytes(TH), L.c_plane_pieces(TH)
    assert plane_bytes % 1024 == 0 and pieces * 1024 == plane_bytes
    buf4, buf3 = (ctypes.c_int * 4)(), (ctypes.c_int * 3)()
    dma = np.zeros((pieces, 64, 4), np.int64)
    for pc in range(pieces):
        for ln in range(64):
            L.c_dma_lane(pc, ln, buf4)
            dma[pc, ln] = list(buf4)
    # every byte of a plane is written exactly once per patch load
    offs = np.sort(dma[:, :, 3].ravel())
    assert np.array_equal(offs, np.arange(0, plane_bytes, 16))
    nfrag, b_passes = bn // 16, (bn + 63) // 64
    wl = np.zeros((b_passes, 4, 64, 3), np.int64)
    for j in range(b_passes):
        for wv in range(4):
            for ln in range(64):
                L.c_w_lane(j, wv, ln, buf3)
                wl[j, wv, ln] = list(buf3)
    tiles_n, tiles_x, tiles_y = cout_pad // bn, (W + 15) // 16, (H + TH - 1) // TH
    total = N * tiles_y * tiles_x * tiles_n
    out = np.full((N, H, W, cout_pad), np.nan, np.float32)
    lanes = range(64)
    foff = np.zeros((4 * mi, 3, 64), np.int64)     # [tile row][tap column q][lane]: filter row 0
    for oy in range(4 * mi):
        for q in range(3):
            for ln in lanes:
                foff[oy, q, ln] = L.c_frag_offset(oy, ln & 15, 0, q, ln >> 4)
    wfoff = np.array([[L.c_w_frag_offset(jf, ln) for ln in lanes] for jf in range(nfrag)])
    row_step = L.c_row_step()
    for tile in range(total):
        L.c_tile_of(tile, tiles_n, tiles_x, tiles_y, TH, bn, buf4)
        n, y0, x0, n0 = list(buf4)
        # ---- patch planes (the kernel: hi = y0 - 1 + py, wi = x0 - 1 + px; zero fill outside the image, in the pitch padding and beyond Cin)
        patch = np.full((KC, plane_bytes // 2), np.float16(np.nan))   # NaN: a fragment read of a byte nobody wrote would show
        for kc in range(KC):
            for pc in range(pieces):
                for ln in range(64):
                    py, px_, clog, off = (int(v) for v in dma[pc, ln])
                    hi, wi, c0 = y0 - 1 + py, x0 - 1 + px_, kc * 32 + clog * 8
                    ok = px_ < 18 and 0 <= hi < H and 0 <= wi < W and c0 < Cin
                    patch[kc, off // 2:off // 2 + 8] = x[n, hi, wi, c0:c0 + 8] if ok else 0
        assert not np.isnan(patch.astype(np.float32)).any()
        acc = np.zeros((4, mi, nfrag, 16, 16), np.float32)   # [wave][fragment][column fragment][channel][pixel]
        for r in range(3):
            for q in range(3):
                for kc in range(KC):
                    e = (r * 3 + q) * KC + kc
                    stage = np.full((b_passes * 64 * 32,), np.float16(np.nan))
                    for j in range(b_passes):
                        for wv in range(4):
                            for ln in range(64):
                                row, clog, off = (int(v) for v in wl[j, wv, ln])
                                stage[off // 2:off // 2 + 8] = packed[n0 + row, e * 32 + clog * 8:e * 32 + clog * 8 + 8] if row < bn else 0
                    for wv in range(4):
                        for i in range(mi):
                            A = np.zeros((16, 32), np.float32)   # B operand of the MFMA: pixel (lane & 15), k chunk (lane >> 4)
                            for ln in lanes:
                                o = int(foff[wv * mi + i, q, ln]) + r * row_step
                                A[ln & 15, (ln >> 4) * 8:(ln >> 4) * 8 + 8] = patch[kc, o // 2:o // 2 + 8]
                            for jf in range(nfrag):
                                Bm = np.zeros((16, 32), np.float32)   # A operand: channel (lane & 15) of column fragment jf
                                for ln in lanes:
                                    o = int(wfoff[jf, ln])
                                    Bm[ln & 15, (ln >> 4) * 8:(ln >> 4) * 8 + 8] = stage[o // 2:o // 2 + 8]
                                acc[wv, i, jf] += Bm @ A.T
        assert not np.isnan(acc).any()
        # epilogue's pixel_of: row t of the tile = 16 * tile row + column
        for wv in range(4):
            for i in range(mi):
                y = y0 + wv * mi + i
                for pxl in range(16):
                    xx = x0 + pxl
                    if y < H and xx < W:
                        for jf in range(nfrag):
                            out[n, y, xx, n0 + jf * 16:n0 + jf * 16 + 16] = acc[wv, i, jf, :, pxl]
    assert not np.isnan(out[..., :Cout]).any(), "an output pixel no tile wrote"
    ref = _conv_ref(x.astype(np.float64), w)
    err = np.abs(out[..., :Cout] - ref).max()
    assert err < 2e-3 * max(1.0, np.abs(ref).max()), err
    assert np.all(out[..., Cout:] == 0)
