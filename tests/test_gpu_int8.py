"""f3: INT8 path.  The int8 MFMA implicit-GEMM conv (v_mfma_i32_16x16x64_i8, int32 accumulation is exact) against an integer
reference; calibration and engine tests further down."""
import numpy as np
import pytest

from tensorrtx_amd import capi

I8_CASES = [
    # N, H, W, Cin, Cout, k, s, p, act1, residual ("i8" | "f16" | None), act2, int8 output
    (2, 20, 20, 64, 64, 3, 1, 1, "silu", None, "none", True),
    (2, 20, 20, 64, 64, 3, 1, 1, "silu", None, "none", False),
    (1, 40, 40, 32, 64, 1, 1, 0, "silu", None, "none", True),     # Cin 32 inside a 64-wide k-step
    (2, 17, 13, 128, 128, 3, 2, 1, "relu", None, "none", True),    # stride 2, ragged M
    (2, 14, 14, 256, 64, 1, 1, 0, "none", "i8", "relu", True),     # resnet tail: relu(conv + int8 shortcut), requantised
    (1, 20, 20, 64, 80, 3, 1, 1, "silu", "f16", "none", False),    # 5 column fragments, fp16 residual, fp16 out
    (3, 80, 80, 64, 64, 3, 1, 1, "silu", None, "none", True),      # many tiles
    (1, 10, 10, 512, 256, 1, 1, 0, "silu", None, "none", True),    # long K
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", I8_CASES)
def test_conv_i8_mfma_vs_integer_reference(gpu, case):
    import torch
    import torch.nn.functional as F
    N, H, W, Cin, Cout, k, s, p, act1, res_kind, act2, out_i8 = case
    g = torch.Generator().manual_seed(abs(hash(case)) & 0xFFFF)
    xq = torch.randint(-127, 128, (N, H, W, Cin), generator=g, dtype=torch.int32)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    s_in = 0.02
    packed, wscale = capi.pack_conv_weights_i8(w.numpy())
    cout_pad = packed.shape[0]
    # the quantised weights the kernel multiplies with, recomputed independently
    amax = w.abs().reshape(Cout, -1).max(1).values
    sw = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
    wq = torch.clamp(torch.round(w / sw[:, None, None, None]), -127, 127)
    assert np.allclose(wscale[:Cout], sw.numpy(), rtol=1e-6)
    acc = F.conv2d(xq.permute(0, 3, 1, 2).double(), wq.double(), None, stride=s, padding=p)  # exact integers in float64
    cscale = torch.zeros(cout_pad)
    cscale[:Cout] = s_in * torch.from_numpy(wscale[:Cout])
    bias_pad = torch.zeros(cout_pad)
    bias_pad[:Cout] = bias
    y = acc.float() * cscale[:Cout, None, None] + bias[:, None, None]
    act = {"none": lambda t: t, "relu": torch.relu, "silu": F.silu}
    y = act[act1](y)
    Ho, Wo = y.shape[2:]
    res = None
    res_scale = 0.05
    if res_kind == "i8":
        res = torch.randint(-127, 128, (N, Ho, Wo, Cout), generator=g, dtype=torch.int32).to(torch.int8)
        y = y.half().float() + res.float().permute(0, 3, 1, 2) * res_scale
    elif res_kind == "f16":
        res = torch.randn(N, Ho, Wo, Cout, generator=g).half()
        y = y.half().float() + res.float().permute(0, 3, 1, 2)
    y = act[act2](y).permute(0, 2, 3, 1)
    s_out = float(y.abs().max()) / 127.0
    got = capi.conv2d_nhwc_i8(xq.to(torch.int8).to(gpu), torch.from_numpy(packed).to(gpu), cscale.to(gpu), bias_pad.to(gpu), Cout, k, k, s, p, act1,
                              out_scale=s_out if out_i8 else None, residual=res.to(gpu) if res is not None else None, res_scale=res_scale, act2=act2)
    torch.cuda.synchronize()
    if out_i8:
        want = torch.clamp(torch.round(y.half().float() / s_out), -127, 127)
        d = (got.cpu().float() - want).abs()
        assert d.max().item() <= 1 and (d > 0).float().mean().item() < 0.02   # fp32 contraction / fp16 rounding may move a value across .5
    else:
        err = (got.cpu().float() - y).abs().max().item()
        assert err <= 2e-3 * max(float(y.abs().max()), 1.0) + 1e-3
