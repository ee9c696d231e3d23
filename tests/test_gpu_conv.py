"""GPU numerics: fused implicit-GEMM MFMA convolution vs a plain PyTorch fp32 reference of the same op
(conv2d on the fp16-rounded operands, fp32 math).  Shapes are taken from the YOLOv8n / ResNet-50
inventories."""
import os

import numpy as np
import pytest

from tensorrtx_amd import capi

pytestmark = pytest.mark.gpu


def _ref(x_nhwc_h, w, bias, stride, pad, act1, res_h, act2):
    import torch
    import torch.nn.functional as F
    x = x_nhwc_h.float().permute(0, 3, 1, 2)
    y = F.conv2d(x, w.half().float(), bias, stride=stride, padding=pad)
    act = {"none": lambda t: t, "relu": torch.relu, "silu": F.silu, "sigmoid": torch.sigmoid,
           "leaky": lambda t: F.leaky_relu(t, 0.1)}
    y = act[act1](y)
    if res_h is not None:
        y = y.half().float() + res_h.float().permute(0, 3, 1, 2)  # kernel rounds to fp16 before the add
    y = act[act2](y)
    return y.permute(0, 2, 3, 1).contiguous()


CASES = [
    # N, H, W, Cin, Cout, k, s, p, act1, residual, act2
    (2, 40, 40, 8, 16, 3, 2, 1, "silu", False, "none"),      # stem-like (Cin padded to 8, K=72 -> Kpad 96)
    (2, 32, 32, 16, 32, 3, 2, 1, "silu", False, "none"),
    (2, 20, 20, 32, 32, 1, 1, 0, "silu", False, "none"),
    (2, 20, 20, 16, 16, 3, 1, 1, "silu", True, "none"),      # C2F bottleneck with shortcut
    (1, 20, 20, 64, 80, 3, 1, 1, "silu", False, "none"),     # cv3 branch, Cout = 80 (5 fragments)
    (1, 20, 20, 80, 80, 3, 1, 1, "silu", False, "none"),     # Cin = 80 (tap boundary inside a k-tile)
    (1, 20, 20, 80, 80, 1, 1, 0, "none", False, "none"),     # detect head 1x1 with bias
    (2, 10, 10, 128, 256, 3, 2, 1, "silu", False, "none"),
    (1, 10, 10, 384, 256, 1, 1, 0, "silu", False, "none"),
    (1, 14, 14, 256, 64, 1, 1, 0, "none", True, "relu"),     # resnet bottleneck tail: relu(conv + shortcut)
    (1, 17, 13, 48, 64, 3, 1, 1, "relu", False, "none"),     # ragged M (221 pixels), Cin 48
    (1, 9, 9, 24, 40, 5, 1, 2, "leaky", False, "none"),      # 5x5, Cout 40 (padded to 48 -> bn 16)
    # --- shapes served by the weight-stationary persistent kernel (conv_ws.hip): several tiles per workgroup, image borders,
    # ragged last tiles, every wave split (WC 1/2/4), both tile sizes, row-aligned and whole-row tile geometry
    (3, 80, 80, 64, 64, 3, 1, 1, "silu", False, "none"),     # 16-px row tiles (ROWS), WC=2, 150 tiles
    (3, 80, 80, 32, 32, 3, 1, 1, "silu", True, "none"),      # KC=1, residual
    (2, 40, 40, 64, 64, 3, 1, 1, "silu", True, "none"),      # W=40: whole-row tiles 3x40, fragments cross rows
    (5, 20, 20, 64, 64, 3, 1, 1, "silu", False, "none"),     # W=20: 64-pixel tiles
    (2, 80, 80, 64, 80, 3, 1, 1, "silu", False, "none"),     # Cout 80 on 8 column fragments (WC=4)
    (2, 40, 40, 128, 64, 3, 1, 1, "silu", False, "none"),    # KC=4, one column fragment per wave
    (2, 23, 37, 64, 64, 3, 1, 1, "relu", False, "none"),     # odd map: partial tiles in both directions
    (4, 160, 160, 32, 32, 1, 1, 0, "silu", False, "none"),   # 1x1 streaming, 800 tiles
    (2, 160, 160, 48, 32, 1, 1, 0, "silu", False, "none"),   # Cin 48 in a 64-wide k-slice (zero-filled chunk tail)
    (2, 80, 80, 128, 64, 1, 1, 0, "silu", False, "none"),
    (2, 80, 80, 192, 64, 1, 1, 0, "silu", False, "none"),    # KC=6
    (2, 80, 80, 96, 64, 1, 1, 0, "silu", False, "none"),     # KC=3
    (3, 40, 40, 256, 128, 1, 1, 0, "silu", False, "none"),   # KC=8, WC=2
    (3, 40, 40, 384, 128, 1, 1, 0, "silu", False, "none"),   # KC=12, WC=4, 64-pixel tiles
    (3, 40, 40, 192, 128, 1, 1, 0, "silu", False, "none"),
    (7, 20, 20, 256, 256, 1, 1, 0, "silu", False, "none"),   # WC=4, NFW=4
    (3, 80, 80, 80, 80, 1, 1, 0, "none", False, "none"),     # head 1x1, 5 column fragments, bias only
    (1, 7, 9, 64, 64, 1, 1, 0, "none", True, "relu"),        # 63 pixels: a single ragged tile, residual + relu
]


@pytest.mark.parametrize("case", CASES)
def test_conv_igemm_vs_torch(gpu, case):
    import torch
    N, H, W, Cin, Cout, k, s, p, act1, use_res, act2 = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = (torch.randn(N, H, W, Cin, generator=g) * 1.0).half()
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = (torch.randn(N, Ho, Wo, Cout, generator=g)).half() if use_res else None
    packed, cout_pad, kpad, bn = capi.pack_conv_weights_f16(w.numpy(), cin_pad=Cin)
    bias_pad = torch.zeros(cout_pad)
    bias_pad[:Cout] = bias
    y = capi.conv2d_nhwc_f16(x.to(gpu), torch.from_numpy(packed.view(np.int16)).to(gpu), bias_pad.to(gpu), Cout, k, k,
                             s, p, act1, res.to(gpu) if use_res else None, act2)
    torch.cuda.synchronize()
    ref = _ref(x, w, bias, s, p, act1, res, act2)
    got = y.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-3 * max(scale, 1.0) + 1e-3, f"max err {err} (scale {scale})"


def test_conv_writes_channel_slice_of_wider_buffer(gpu):
    """Concat fusion: the conv stores into channels [16, 48) of a 64-channel NHWC buffer and reads a
    32-channel slice of a 64-channel input (strided views, block.cpp:134-149 C2F without copies)."""
    import torch
    g = torch.Generator().manual_seed(3)
    xin = torch.randn(2, 12, 12, 64, generator=g).half()
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.06
    packed, cout_pad, kpad, bn = capi.pack_conv_weights_f16(w.numpy(), cin_pad=32)
    buf = torch.full((2, 12, 12, 64), 7.0).half().to(gpu)
    xg = xin.to(gpu)
    x_view = xg[..., 32:]            # channel slice, stride(2) stays 64
    out_view = buf[..., 16:48]
    capi.conv2d_nhwc_f16(x_view, torch.from_numpy(packed.view(np.int16)).to(gpu), None, 32, 3, 3, 1, 1, "none",
                         out=out_view, out_ld=64)
    torch.cuda.synchronize()
    ref = _ref(xin[..., 32:].contiguous(), w, None, 1, 1, "none", None, "none")
    got = buf.float().cpu()
    assert (got[..., :16] == 7.0).all() and (got[..., 48:] == 7.0).all()
    assert (got[..., 16:48] - ref).abs().max().item() < 5e-3


# Every launch configuration the tactic tuner may pick for a layer (runtime/tune.cpp, conv_tactics): all of them must be the same
# convolution.  Tile shapes (column-tile width, 64- / 128-row tiles, 32- / 64-wide k-steps) accumulate over K in the same order:
# bit-identical.  The wave-split-K and weight-stationary kernels sum in a different order: within the fp16 tolerance.
TACTIC_CASES = [
    (2, 20, 20, 128, 128, 3, 1, 1, "silu", True, "none"),    # 13 tactics: 3 tile widths x 2 heights x 2 k-steps + wave-split-K
    (2, 40, 40, 64, 64, 3, 1, 1, "silu", False, "none"),
    (1, 20, 20, 64, 80, 3, 1, 1, "silu", False, "none"),     # Cout 80: 5 column fragments, odd item count in the 64-row epilogue
    (2, 20, 20, 256, 256, 1, 1, 0, "silu", False, "none"),   # 1x1 with 64-wide k-steps
    (3, 80, 80, 32, 32, 3, 1, 1, "silu", True, "none"),      # weight-stationary default vs the implicit-GEMM tiles
    (1, 17, 13, 64, 64, 3, 1, 1, "relu", False, "none"),     # ragged M (221 pixels): partial 64- and 128-row tiles
    (2, 10, 10, 128, 256, 3, 2, 1, "silu", False, "none"),   # stride 2
    (1, 14, 14, 256, 64, 1, 1, 0, "none", True, "relu"),     # residual + second activation
    (1, 9, 9, 24, 40, 5, 1, 2, "leaky", False, "none"),      # Cin 24 (one ragged k-chunk), Cout 40 -> 48
    (2, 20, 20, 80, 80, 3, 1, 1, "silu", False, "none"),     # Cin 80: a ragged third channel slice per filter row (row-reuse kernel)
    (3, 7, 5, 64, 128, 3, 1, 1, "none", False, "none"),      # tiny map: padding columns are 2 of 7, tiles straddle rows and images
    (6, 160, 160, 64, 64, 3, 1, 1, "silu", True, "none"),    # 600 tiles of 256 rows: the 256-row tile joins the candidates
    (8, 160, 160, 48, 32, 1, 1, 0, "silu", False, "none"),   # 1x1 over a large map, 32-wide column tiles, Cin 48 in a 64-wide slice
    (16, 57, 55, 256, 256, 1, 1, 0, "relu", True, "relu"),   # the large-GEMM tile (256 x 128, 2 x 2 waves) joins: 1x1, ragged last tile
    (9, 56, 56, 128, 384, 3, 1, 1, "relu", False, "none"),   # ... and on a 3x3 (K = 1152), three column tiles
    # round 4: the 256 x 256 x 64 role-alternating tile (conv_gemm256.hip) joins for large plain GEMMs: K >= 512, Cout a multiple of 256
    (41, 40, 40, 512, 512, 1, 1, 0, "relu", True, "relu"),   # res5-like 512 -> 512 + shortcut, ragged last row tile (65 600 = 256 x 256 + 64)
    (24, 56, 49, 1024, 256, 1, 1, 0, "silu", False, "none"),  # K = 1024, one column tile
    (11, 57, 55, 2048, 512, 1, 1, 0, "none", False, "relu"),  # K = 2048 (res5's 2048 -> 512), odd tile count per XCD
    (11, 56, 56, 512, 512, 3, 1, 1, "relu", False, "none"),   # ... and its im2col form on a 3x3 (res5's 512 -> 512, K = 4608): borders, images, ragged last tile
    (140, 23, 21, 256, 256, 3, 1, 1, "silu", True, "none"),    # small maps: most rows touch a border, tiles straddle several images
    # round 6: the resident-operand 3x3 kernel (conv_res.hip, ws == 7) joins for 32 -> 32, 64 -> 64, 64 -> 80: persistent workgroups with several tiles each
    (32, 40, 40, 64, 64, 3, 1, 1, "silu", True, "relu"),      # 480 tiles on 240 workgroups: both halves busy, shortcut + second activation
    (16, 80, 80, 64, 80, 3, 1, 1, "silu", False, "none"),     # Cout 80: the unpaired fifth fragment's 8-byte stores, 3-4 tiles per workgroup
    (9, 83, 77, 32, 32, 3, 1, 1, "none", True, "none"),       # 16-row tiles, ragged in both directions, two workgroups per CU, shortcut without activation
    (5, 13, 9, 64, 64, 3, 1, 1, "relu", False, "none"),       # fewer tiles than workgroup slots: one tile per workgroup, the second half idle
    # ... and its 1x1 sibling (ws == 8): 16 independent waves per persistent workgroup, the column tile's weights resident, A straight into registers
    (32, 80, 80, 128, 64, 1, 1, 0, "silu", False, "none"),    # 12 800 row fragments, four chunks of k-steps... one chunk of four
    (32, 40, 40, 192, 128, 1, 1, 0, "silu", True, "relu"),    # six k-steps: a short second chunk; 128-wide column tile, shortcut + second activation
    (32, 20, 20, 512, 256, 1, 1, 0, "silu", False, "none"),   # K = 512: four chunks, four column tiles of 64 (64 KB of weights each)
    (7, 33, 29, 96, 80, 1, 1, 0, "none", True, "none"),       # three k-steps (odd chunk count: 1), Cout 80: the unpaired fragment, ragged last row fragment
    (3, 57, 55, 48, 32, 1, 1, 0, "relu", False, "none"),      # Cin 48 in a 64-wide K: the ragged chunk's lanes are range-checked to zero
    # ... and its thin 3x3 form (TAPS2): 16 input channels, two filter taps per k-step, A straight into registers from the tap-shifted pixels
    (8, 160, 160, 16, 16, 3, 1, 1, "silu", True, "none"),     # YOLOv8n model.2.m.0.cv2 (+ shortcut): one column fragment, borders on every side, many fragments per wave
    (5, 50, 64, 16, 32, 3, 2, 1, "silu", False, "none"),      # stride 2 (model.1: 16 -> 32): output rows of 32 pixels, odd output height
    (3, 21, 48, 16, 16, 3, 1, 1, "none", False, "relu"),      # fewer fragments than waves on most workgroups
]


def test_the_256x256_tile_chosen_at_the_build_batch_runs_at_a_smaller_batch(gpu):
    """conv_gemm256_possible() held the '>= 96 tiles' profitability rule, which depends on the RUNTIME batch - a layer the tuner had put on the
    256 x 256 x 64 tile at max_batch was refused at a smaller batch and the whole enqueue failed (R-CNN res4 at 800 x 1344: 132 tiles at batch 2, 68 at batch 1).
    The rule now only decides candidacy; the tile itself launches at any batch and returns the 128-row tiles' bits."""
    import torch
    g = torch.Generator().manual_seed(5)
    Cin, Cout, H, W = 256, 256, 50, 84      # batch 8: 132 tiles (a candidate), batch 1: 17 tiles
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5
    packed, cout_pad, kpad, bn = capi.pack_conv_weights_f16(w.numpy(), cin_pad=Cin)
    wg, bg = torch.from_numpy(packed.view(np.int16)).to(gpu), torch.zeros(cout_pad, device=gpu)
    t256 = (256, 64, 256, 1, 1, 0)
    assert t256 in capi.conv2d_tactics(8, H, W, Cin, Cout, 1, 1, 0) and t256 not in capi.conv2d_tactics(1, H, W, Cin, Cout, 1, 1, 0)
    for n in (8, 1):
        x = torch.randn(n, H, W, Cin, generator=g).half().to(gpu)
        try:
            capi.conv_force_tactic(t256)
            y = capi.conv2d_nhwc_f16(x, wg, bg, Cout, 1, 1, 1, 0, "relu")
            capi.conv_force_tactic((128, 32, 128, 1, 1, 0))
            y0 = capi.conv2d_nhwc_f16(x, wg, bg, Cout, 1, 1, 1, 0, "relu")
        finally:
            capi.conv_force_tactic(None)
        torch.cuda.synchronize()
        assert torch.equal(y, y0), f"batch {n}"


@pytest.mark.parametrize("case", TACTIC_CASES)
def test_every_conv_tactic_is_the_same_convolution(gpu, case):
    import torch
    N, H, W, Cin, Cout, k, s, p, act1, use_res, act2 = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, H, W, Cin, generator=g).half()
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = torch.randn(N, Ho, Wo, Cout, generator=g).half() if use_res else None
    packed, cout_pad, kpad, bn = capi.pack_conv_weights_f16(w.numpy(), cin_pad=Cin)
    bias_pad = torch.zeros(cout_pad)
    bias_pad[:Cout] = bias
    xg, wg, bg = x.to(gpu), torch.from_numpy(packed.view(np.int16)).to(gpu), bias_pad.to(gpu)
    rg = res.to(gpu) if use_res else None
    ref = _ref(x, w, bias, s, p, act1, res, act2)
    scale = ref.abs().max().item()
    tactics = capi.conv2d_tactics(N, H, W, Cin, Cout, k, s, p, residual=use_res)
    assert len(tactics) >= 2 and len(set(tactics)) == len(tactics)
    if os.environ.get("TRTX_BIG_VARIANT") and Cout % 128 == 0 and Cin % 64 == 0 and N * Ho * Wo * (Cout // 128) >= 256 * 256:
        assert (128, 64, 256, 1, 1, 0) in tactics   # the large-GEMM configurations (experiments: tools/gemm_tactics.py)
    if Cin % 64 == 0 and Cout % 256 == 0 and s == 1 and k * k * Cin >= 256 and (k == 3 or (k == 1 and p == 0)) and ((N * Ho * Wo + 255) // 256) * (Cout // 256) >= 96:
        assert (256, 64, 256, 1, 1, 0) in tactics   # conv_gemm256_possible
    if Cin == 16 and k == 3 and p == 1 and Wo % 16 == 0:
        assert any(t[4] == 8 for t in tactics), "the A-direct kernel's thin 3x3 form is a candidate for 16-channel 3x3 layers with whole 16-pixel output rows"
    exact = None
    try:
        for t in tactics:
            capi.conv_force_tactic(t)
            y = capi.conv2d_nhwc_f16(xg, wg, bg, Cout, k, k, s, p, act1, rg, act2)
            torch.cuda.synchronize()
            got = y.float().cpu()
            err = (got - ref).abs().max().item()
            assert err <= 2e-3 * max(scale, 1.0) + 1e-3, f"tactic {t}: max err {err} (scale {scale})"
            if t[3] == 1 and t[4] in (1, 3, 7, 8) and t[5] == 0:  # plain implicit-GEMM tiles (ws 3 / 7 / 8: the resident-patch and the resident-operand 3x3 / 1x1 kernels walk K in the same order)
                if exact is None:
                    exact = got
                else:
                    assert torch.equal(got, exact), f"tactic {t} is not bit-identical to the other tile shapes"
    finally:
        capi.conv_force_tactic(None)
