"""GPU numerics of the fp32 engines' convolution (kernels/conv_igemm_f32.hip: implicit GEMM on v_mfma_f32_16x16x4_f32) against a plain PyTorch fp32
reference of the same op, through the C ABI (trtx_op_conv2d_nhwc_f32).  The tolerance is fp32 round-off of a K-long sum: 2e-6 * sqrt(K) of the output
scale - three orders of magnitude inside what BASELINE asks of the fp32 build (1e-4 on logits of O(10)) - and every tile shape of a layer must return
the SAME bits (one summation order per layer whatever the tactic).  Shapes: yolov8/src/model.cpp:115-251 and resnet/resnet50.cpp:165-206 (SURVEY
Appendix C), plus the ragged / sliced cases the fp16 kernel is tested with."""
import numpy as np
import pytest

from tensorrtx_amd import capi

pytestmark = pytest.mark.gpu


def _ref(x_nhwc, w, bias, stride, pad, act1, res, act2):
    import torch
    import torch.nn.functional as F
    x = x_nhwc.double().permute(0, 3, 1, 2)
    y = F.conv2d(x, w.double(), bias.double(), stride=stride, padding=pad)
    act = {"none": lambda t: t, "relu": torch.relu, "silu": F.silu, "sigmoid": torch.sigmoid, "leaky": lambda t: F.leaky_relu(t, 0.1)}
    y = act[act1](y)
    if res is not None:
        y = y + res.double().permute(0, 3, 1, 2)
    y = act[act2](y)
    return y.permute(0, 2, 3, 1).contiguous()


CASES = [
    # N, H, W, Cin, Cout, k, s, p, act1, residual, act2
    (2, 40, 40, 4, 16, 3, 2, 1, "silu", False, "none"),      # stem: 3 channels padded to 4, two filter taps per 16-float k-step
    (1, 30, 34, 4, 64, 7, 2, 3, "relu", False, "none"),      # ResNet stem 7x7 / 2 (49 taps: row and column masks)
    (2, 32, 32, 16, 32, 3, 2, 1, "silu", False, "none"),     # Cin 16 = exactly one step per tap
    (2, 20, 20, 32, 32, 1, 1, 0, "silu", False, "none"),     # plain GEMM
    (2, 20, 20, 16, 16, 3, 1, 1, "silu", True, "none"),      # C2f bottleneck with shortcut, one column fragment
    (1, 20, 20, 64, 80, 3, 1, 1, "silu", False, "none"),     # Cout 80 (5 fragments)
    (1, 20, 20, 80, 80, 3, 1, 1, "silu", False, "none"),     # Cin 80 = 5 steps per tap
    (1, 20, 20, 80, 80, 1, 1, 0, "none", False, "none"),     # detect head 1x1 with bias
    (2, 10, 10, 128, 256, 3, 2, 1, "silu", False, "none"),
    (1, 10, 10, 384, 256, 1, 1, 0, "silu", False, "none"),
    (1, 14, 14, 256, 64, 1, 1, 0, "none", True, "relu"),     # ResNet bottleneck tail: relu(conv + shortcut)
    (1, 17, 13, 48, 64, 3, 1, 1, "relu", False, "none"),     # ragged M (221 pixels), Cin 48
    (1, 9, 9, 24, 40, 5, 1, 2, "leaky", False, "none"),      # 5x5, Cin 24 -> 32 per tap (zero-filled chunk tail), Cout 40 -> 48
    (1, 9, 9, 20, 21, 3, 1, 1, "sigmoid", False, "none"),    # Cout 21: element-wise stores
    (3, 80, 80, 64, 64, 3, 1, 1, "silu", False, "none"),     # 150 row tiles of 128
    (2, 23, 37, 64, 64, 3, 1, 1, "relu", True, "relu"),      # odd map, residual + relu
    (4, 160, 160, 32, 32, 1, 1, 0, "silu", False, "none"),   # 800 row tiles, plain GEMM, two steps
    (2, 80, 80, 192, 64, 1, 1, 0, "silu", False, "none"),
    (7, 20, 20, 256, 256, 1, 1, 0, "silu", False, "none"),
    (1, 7, 9, 64, 64, 1, 1, 0, "none", True, "relu"),        # 63 pixels: a single ragged tile
    (2, 40, 40, 32, 32, 3, 1, 1, "silu", False, "none"),     # resident patch: two planes of a 16-row patch
    (1, 20, 20, 128, 64, 3, 1, 1, "silu", False, "none"),    # resident patch: eight planes (one workgroup per CU)
    (1, 20, 20, 128, 128, 3, 1, 1, "silu", True, "none"),    # ... two 64-wide column tiles over one patch; wave roles at 128 columns
    (16, 80, 80, 32, 32, 3, 1, 1, "silu", True, "none"),     # round 6, the resident-operand kernel with fp32 operands: several tiles per persistent workgroup, shortcut
    (9, 43, 37, 64, 64, 3, 1, 1, "leaky", False, "relu"),    # ... 64 -> 64 as 2 x 32 and as 4 x 16 columns (column tiles bound to workgroups), ragged map, slow activation
    (16, 40, 40, 128, 64, 1, 1, 0, "silu", True, "none"),    # ... its 1x1 form: two chunks of four k-steps, many row fragments per wave, shortcut
    (3, 33, 31, 48, 80, 1, 1, 0, "leaky", False, "none"),    # ... a short chunk (three k-steps), five column fragments, ragged M, slow activation
]


def _run(case, gpu, tile=None):
    import torch
    N, H, W, Cin, Cout, k, s, p, act1, use_res, act2 = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = torch.randn(N, Ho, Wo, Cout, generator=g) if use_res else None
    packed, cout_pad, kpad, cink = capi.pack_conv_weights_f32(w.numpy(), cin_pad=Cin)
    bias_pad = torch.zeros(cout_pad)
    bias_pad[:Cout] = bias
    y = capi.conv2d_nhwc_f32(x.to(gpu), torch.from_numpy(packed).to(gpu), bias_pad.to(gpu), Cout, k, k, s, p, act1, res.to(gpu) if use_res else None, act2,
                             tile=tile)
    torch.cuda.synchronize()
    return y.cpu(), (x, w, bias, s, p, act1, res, act2), Cin * k * k


@pytest.mark.parametrize("case", CASES)
def test_conv_f32_mfma_vs_torch_and_every_tile_shape_is_the_same_bits(gpu, case):
    import torch
    got, ref_args, K = _run(case, gpu)
    ref = _ref(*ref_args)
    err = (got.double() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1.0)
    tol = 2e-6 * (K ** 0.5) * scale
    assert err <= tol, f"max err {err} beyond {tol} (K {K}, scale {scale})"
    N, H, W, Cin, Cout, k, s, p, _, use_res, _ = case
    tiles = capi.conv2d_tactics_f32(N, H, W, Cin, Cout, k, s, p, residual=use_res)
    assert len(tiles) >= 1
    kinds = {t[2] for t in tiles}
    if k == 3 and s == 1 and p == 1 and Cin % 16 == 0 and Cin // 16 in (1, 2, 3, 4, 5, 8) and Cout % 4 == 0:
        assert 3 in kinds, "the resident-patch kernel is a candidate for every 3x3 stride-1 layer it can take"
    if k == 3 and s == 1 and p == 1 and Cin in (16, 32, 64, 80) and Cout % 4 == 0:
        assert 7 in kinds, "the resident-operand 3x3 kernel (conv_res.hip, fp32 operands) is a candidate for the 16- to 80-channel 3x3 stride-1 layers"
    if k == 1 and s == 1 and Cin % 16 == 0 and Cout % 4 == 0 and Cin * 16 * 4 <= 98304:
        assert 8 in kinds, "the resident-weights 1x1 kernel (conv_res.hip, fp32 operands) is a candidate for every 1x1 layer whose narrowest column tile fits LDS"
    if Cin > 8 and ((Cout + 15) // 16 * 16) % 32 == 0:
        assert 6 in kinds and 5 in kinds, "wave roles / register-staged operands are candidates wherever a 32-wide column tile exists"
    for t in tiles[1:]:
        other, _, _ = _run(case, gpu, tile=t)
        assert torch.equal(other, got), f"tile {t} differs from tile {tiles[0]}"


def test_conv_f32_channel_slices_of_wider_buffers(gpu):
    """Concat without copies (block.cpp:134-149): reads channels [32, 64) of a 64-channel tensor, writes channels [16, 48) of another; the rest stays."""
    import torch
    g = torch.Generator().manual_seed(3)
    xin = torch.randn(2, 12, 12, 64, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.06
    packed, cout_pad, _, _ = capi.pack_conv_weights_f32(w.numpy(), cin_pad=32)
    buf = torch.full((2, 12, 12, 64), 7.0).to(gpu)
    xg = xin.to(gpu)
    capi.conv2d_nhwc_f32(xg[..., 32:], torch.from_numpy(packed).to(gpu), None, 32, 3, 3, 1, 1, "none", out=buf[..., 16:48], out_ld=64)
    torch.cuda.synchronize()
    out = buf.cpu()
    ref = _ref(xin[..., 32:], w, torch.zeros(32), 1, 1, "none", None, "none")
    assert (out[..., 16:48].double() - ref).abs().max().item() < 1e-5
    assert torch.all(out[..., :16] == 7.0) and torch.all(out[..., 48:] == 7.0)


def test_conv_f32_is_an_fmaf_chain_exact_on_small_integers(gpu):
    """Integer-valued operands whose sums stay below 2^24: every product and partial sum is exact in fp32, so the MFMA result must equal the integer
    convolution exactly, in any summation order - the fragment / k-index mapping of compute() has no slack to hide in."""
    import torch
    g = torch.Generator().manual_seed(11)
    x = torch.randint(-8, 9, (2, 19, 21, 48), generator=g).float()
    w = torch.randint(-8, 9, (80, 48, 3, 3), generator=g).float()
    packed, cout_pad, _, _ = capi.pack_conv_weights_f32(w.numpy(), cin_pad=48)
    y = capi.conv2d_nhwc_f32(x.to(gpu), torch.from_numpy(packed).to(gpu), None, 80, 3, 3, 1, 1)
    torch.cuda.synchronize()
    ref = _ref(x, w, torch.zeros(80), 1, 1, "none", None, "none")
    assert torch.equal(y.cpu().double(), ref)
