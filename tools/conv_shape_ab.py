"""Per-shape A/B of one convolution layer's exchangeable tactics on the GPU (whatever library TRTX_HIP_LIB points at): every tactic conv_tactics() lists
for the shape is forced in turn, timed with events over REPS launches after a warm-up, checked against the first tactic's output (bit-identical where the
summation order is the same).  Default shapes: the 3x3 stride-1 layers of the YOLOv8n b32 step that dominate its LDS-fill bytes (tools/lds_fill_model.py).
    python tools/conv_shape_ab.py [--k 1] [--stride 2] [N H W Cin Cout]..."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tensorrtx_amd import capi  # noqa: E402

SHAPES = [(32, 80, 80, 32, 32), (32, 20, 20, 64, 64), (32, 80, 80, 64, 64), (32, 80, 80, 64, 80), (32, 80, 80, 80, 80), (32, 40, 40, 128, 64), (32, 40, 40, 64, 64), (32, 40, 40, 128, 128),
          (32, 20, 20, 128, 128), (32, 20, 20, 256, 64), (32, 56, 56, 64, 64), (32, 28, 28, 128, 128)]
REPS = 20


K = 3
STRIDE = 1


def one(N, H, W, Cin, Cout):
    gpu = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, Cin, generator=g).half().to(gpu)
    w = torch.randn(Cout, Cin, K, K, generator=g) * (2.0 / (K * K * Cin)) ** 0.5
    packed, cout_pad, kpad, bn = capi.pack_conv_weights_f16(w.numpy(), cin_pad=Cin)
    wg = torch.from_numpy(packed.view(np.int16)).to(gpu)
    bias = torch.zeros(cout_pad, device=gpu)
    flush = torch.empty(256 << 20, dtype=torch.uint8, device=gpu)
    first = None
    print(f"{K}x{K} {Cin} -> {Cout} @ {H}x{W} b{N}: {2 * N * H * W * K * K * Cin * Cout / 1e9:.1f} GFLOP")
    try:
        for t in capi.conv2d_tactics(N, H, W, Cin, Cout, K, STRIDE, K // 2):
            capi.conv_force_tactic(t)
            y = capi.conv2d_nhwc_f16(x, wg, bias, Cout, K, K, STRIDE, K // 2, "silu")
            torch.cuda.synchronize()
            ts = []
            for _ in range(REPS):
                flush.fill_(1)    # the layer's input comes from memory, as after its producer's launch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                capi.conv2d_nhwc_f16(x, wg, bias, Cout, K, K, STRIDE, K // 2, "silu", out=y)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            got = y.float().cpu()
            if first is None:
                first = got
            same = "identical to tactic 0" if torch.equal(got, first) else f"max |diff| to tactic 0 {float((got - first).abs().max()):.3g}"
            kind = "res3" if t[4] == 7 else "res1" if t[4] == 8 else "patch" if t[4] == 3 else ("ws" if t[4] == 2 else ("wsk" if t[3] == 2 else "igemm"))
            print(f"   {kind:6s} bn {t[0]:3d} bk {t[1]:2d} bm {t[2]:3d}   median {sorted(ts)[len(ts) // 2]:7.1f} us  min {min(ts):7.1f}   {same}")
    finally:
        capi.conv_force_tactic(None)


if __name__ == "__main__":
    argv = sys.argv[1:]
    if argv[:1] == ["--k"]:
        K = int(argv[1])
        argv = argv[2:]
    if argv[:1] == ["--stride"]:
        STRIDE = int(argv[1])
        argv = argv[2:]
    a = [int(v) for v in argv]
    shapes = [tuple(a[i:i + 5]) for i in range(0, len(a), 5)] if a else SHAPES
    for s in shapes:
        one(*s)
