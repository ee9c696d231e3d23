"""GPU numerics: the fused convolution-chain kernel (kernels/conv_chain.hip: 3x3 -> 3x3 (+ shortcut), 3x3 -> 3x3 -> 1x1, lone 3x3; the
intermediate tensors stay in LDS) against a plain PyTorch fp32 restatement of the SAME chain run layer by layer with the roundings the
unfused engine performs (fp16 operands, fp32 accumulate, one rounding to fp16 after each activation, shortcut added to the rounded value).
Shapes: the C2f bottlenecks (yolov8/src/block.cpp:98-110) and detect-head arms (yolov8/src/model.cpp:188-251) of YOLOv8n, plus odd maps
(partial tiles in both directions), strided channel slices and every tile / fragment-count instantiation."""
import numpy as np
import pytest

from tensorrtx_amd import capi

pytestmark = pytest.mark.gpu


def _ref_chain(x_h, stages):
    import torch
    import torch.nn.functional as F
    x0 = x_h.float().permute(0, 3, 1, 2)
    y = x0
    for s in stages:
        y = F.conv2d(y, s["w_f"].half().float(), s["b_f"], stride=1, padding=s["k"] // 2)
        y = {"none": lambda t: t, "relu": torch.relu, "silu": F.silu}[s.get("act", "none")](y)
        y = y.half().float()
        if s.get("residual"):
            y = (y + x0).half().float()
    return y.permute(0, 2, 3, 1).contiguous()


def _make(case, gpu, seed):
    import torch
    N, H, W, Cin, spec = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, H, W, Cin, generator=g).half()
    stages, cin = [], Cin
    for (k, cout, act, res) in spec:
        w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        packed = capi.pack_chain_weights_f16(w.numpy())
        stages.append(dict(k=k, cout=cout, act=act, residual=res, w_f=w, b_f=b,
                           w=torch.from_numpy(packed.view(np.int16)).to(gpu), bias=b.to(gpu)))
        cin = cout
    return x, stages


def _check(got, ref, what):
    err = (got - ref).abs()
    tol = 2e-2 + 1e-2 * ref.abs()
    bad = err > tol
    if bad.any():
        idx = bad.nonzero()
        first = idx[0].tolist()
        # where do the wrong values sit?  (pixel rows / columns / channels hit) - the signature of an indexing bug
        ys, xs, cs = sorted(set(idx[:, 1].tolist())), sorted(set(idx[:, 2].tolist())), sorted(set(idx[:, 3].tolist()))
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} values off, max err {float(err.max()):.4f}; first at {first} "
                             f"got {float(got[tuple(first)]):.4f} want {float(ref[tuple(first)]):.4f}; rows {ys[:12]}.. cols {xs[:12]}.. ch {cs[:12]}..")
    return float(err.max())


B3, B1 = 3, 1
CASES = [
    # N, H, W, Cin, [(k, cout, act, residual)...]
    (2, 20, 20, 32, [(3, 32, "silu", False)]),                                              # lone 3x3: the patch kernel itself
    (2, 20, 20, 32, [(3, 32, "silu", False), (3, 32, "silu", True)]),                       # C2f bottleneck with shortcut
    (2, 40, 40, 16, [(3, 16, "silu", False), (3, 16, "silu", True)]),                       # Cin 16: a half-empty 32-channel plane
    (2, 40, 40, 64, [(3, 64, "silu", False), (3, 64, "silu", False)]),                      # head C2f bottleneck (no shortcut)
    (2, 20, 20, 128, [(3, 128, "silu", False), (3, 128, "silu", True)]),                    # 8 column fragments, 4 planes
    (2, 40, 40, 64, [(3, 64, "silu", False), (3, 64, "silu", False), (1, 64, "none", False)]),   # detect-head box arm, level 0 shape
    (2, 40, 40, 64, [(3, 80, "silu", False), (3, 80, "silu", False), (1, 80, "none", False)]),   # class arm: Cout 80 (5 fragments, 2.5 planes)
    (2, 20, 20, 128, [(3, 64, "silu", False), (3, 64, "silu", False), (1, 64, "none", False)]),  # level 1: Cin 128 -> 64
    (1, 20, 20, 256, [(3, 80, "silu", False), (3, 80, "silu", False), (1, 80, "none", False)]),  # level 2: Cin 256 (8 planes) -> 80
    (3, 23, 37, 32, [(3, 32, "relu", False), (3, 32, "relu", True)]),                       # odd map: partial tiles both ways
    (2, 7, 9, 64, [(3, 64, "silu", False), (3, 64, "none", False)]),                        # map smaller than a tile
    (2, 33, 18, 48, [(3, 64, "silu", False), (1, 64, "silu", False)]),                      # Cin 48 (1.5 planes), 3x3 -> 1x1
    (2, 80, 80, 32, [(3, 32, "silu", False), (3, 32, "silu", True)]),                       # 80x80 bottleneck: 16x16 tiles, 6 fragments per wave
    (1, 160, 160, 16, [(3, 16, "silu", False), (3, 16, "silu", True)]),                     # 160x160, 1 column fragment
]


@pytest.mark.parametrize("case", CASES)
def test_chain_vs_layerwise_torch(gpu, case):
    import torch
    x, stages = _make(case, gpu, seed=hash(str(case)) & 0xFFFF)
    capi.poison_lds()   # whatever the kernel reads from LDS without having written it is now NaN
    y = capi.conv_chain_nhwc_f16(x.to(gpu), stages)
    torch.cuda.synchronize()
    _check(y.float().cpu(), _ref_chain(x, stages), str(case))


@pytest.mark.parametrize("tile", [(16, 16), (8, 16), (8, 8), (4, 8), (4, 4)])
@pytest.mark.parametrize("cout", [16, 32, 64, 80])
def test_every_tile_and_fragment_count(gpu, tile, cout):
    """each (column fragments, fragments per wave, ring depth) instantiation that fits, forced through the tile argument"""
    import torch
    case = (2, 21, 35, cout if cout != 80 else 64, [(3, cout, "silu", False), (3, cout, "silu", cout != 80)])
    if capi.conv_chain_plan(case[0], case[1], case[2], case[3], [3, 3], [cout, cout], [0, int(cout != 80)], tile) is None:
        pytest.skip("does not fit")
    x, stages = _make(case, gpu, seed=7)
    capi.poison_lds()
    y = capi.conv_chain_nhwc_f16(x.to(gpu), stages, tile=tile)
    torch.cuda.synchronize()
    _check(y.float().cpu(), _ref_chain(x, stages), f"tile {tile} cout {cout}")


@pytest.mark.parametrize("tile", [(8, 16), (8, 8), (4, 8)])
@pytest.mark.parametrize("cout", [64, 80])
def test_three_stage_arm_at_every_tile(gpu, tile, cout):
    """detect-head arm 3x3 -> 3x3 -> 1x1 at the tiles the launcher uses on 80x80 / 40x40 / 20x20 maps (LDS buffers of the stages alias)"""
    import torch
    case = (2, 27, 45, 64, [(3, cout, "silu", False), (3, cout, "silu", False), (1, cout, "none", False)])
    if capi.conv_chain_plan(case[0], case[1], case[2], case[3], [3, 3, 1], [cout] * 3, None, tile) is None:
        pytest.skip("does not fit")
    x, stages = _make(case, gpu, seed=5)
    capi.poison_lds()
    y = capi.conv_chain_nhwc_f16(x.to(gpu), stages, tile=tile)
    torch.cuda.synchronize()
    _check(y.float().cpu(), _ref_chain(x, stages), f"tile {tile} cout {cout}")


def test_chain_reads_and_writes_channel_slices(gpu):
    """C2f: the bottleneck reads a channel slice of the cv1 output and writes a slice of the concat buffer (block.cpp:134-149)"""
    import torch
    case = (2, 24, 24, 32, [(3, 32, "silu", False), (3, 32, "silu", True)])
    x, stages = _make(case, gpu, seed=3)
    big_in = torch.randn(2, 24, 24, 96).half()
    big_in[..., 32:64] = x
    big_out = torch.full((2, 24, 24, 128), 7.0, dtype=torch.float16, device=gpu)
    din = big_in.to(gpu)
    capi.conv_chain_nhwc_f16(din[..., 32:64], stages, out=big_out[..., 64:96])
    torch.cuda.synchronize()
    got = big_out.float().cpu()
    _check(got[..., 64:96], _ref_chain(x, stages), "slice")
    assert (got[..., :64] == 7.0).all() and (got[..., 96:] == 7.0).all()   # nothing outside the slice is touched


def test_chain_matches_unfused_kernels_bit_for_bit(gpu):
    """the fused chain performs the unfused path's arithmetic in the unfused path's order: same bits as two implicit-GEMM launches"""
    import torch
    case = (2, 40, 40, 64, [(3, 64, "silu", False), (3, 64, "silu", True)])
    x, stages = _make(case, gpu, seed=11)
    xg = x.to(gpu)
    y = capi.conv_chain_nhwc_f16(xg, stages)
    capi.conv_force_tactic((64, 32, 128, 1, 1, 0))   # plain implicit GEMM: no weight-stationary / split-K / row-reuse reordering
    try:
        cur, res = xg, None
        for s in stages:
            packed, cout_pad, kpad, bn = capi.pack_conv_weights_f16(s["w_f"].numpy(), cin_pad=cur.shape[-1])
            bias = torch.zeros(cout_pad)
            bias[:s["cout"]] = s["b_f"]
            cur = capi.conv2d_nhwc_f16(cur, torch.from_numpy(packed.view(np.int16)).to(gpu), bias.to(gpu), s["cout"], 3, 3, 1, 1, s["act"],
                                       xg if s["residual"] else None, "none")
    finally:
        capi.conv_force_tactic(None)
    torch.cuda.synchronize()
    assert torch.equal(y, cur)
