"""Per-shape A/B of the fp32 MFMA convolution's launch configurations on the GPU (kernels/conv_igemm_f32.hip): every (column-tile width, rows per tile,
operand path) conv_tactics_f32() lists for the shape is pinned in turn, timed with events over REPS launches after a warm-up (input flushed out of the caches
in between), checked bit-identical to the first, and priced against the fp32 MFMA peak (157.3 TFLOP/s).  Default shapes: the YOLOv8n b32 layers that carry
its FLOPs.     python tools/conv_f32_shape_ab.py [N H W Cin Cout k stride]... [--act none]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tensorrtx_amd import capi  # noqa: E402

SHAPES = [(32, 80, 80, 64, 64, 3, 1), (32, 80, 80, 80, 80, 3, 1), (32, 40, 40, 128, 64, 3, 1), (32, 40, 40, 64, 64, 3, 1), (32, 20, 20, 128, 128, 3, 1),
          (32, 20, 20, 256, 64, 3, 1), (32, 160, 160, 16, 16, 3, 1), (32, 320, 320, 16, 32, 3, 2), (32, 80, 80, 128, 64, 1, 1), (32, 160, 160, 48, 32, 1, 1),
          (32, 40, 40, 384, 128, 1, 1), (32, 20, 20, 512, 256, 1, 1), (32, 80, 80, 64, 64, 1, 1)]
REPS = 10


ONLY = None
KINDS = None   # --kinds 1,6: only these operand paths (1 LDS-DMA, 5 registers, 6 roles), 16-channel steps


def one(N, H, W, Cin, Cout, k, s, act):
    gpu = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    p = k // 2
    x = torch.randn(N, H, W, Cin, generator=g).to(gpu)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    packed, cout_pad, kpad, cink = capi.pack_conv_weights_f32(w.numpy(), cin_pad=Cin)
    wg = torch.from_numpy(packed).to(gpu)
    bias = torch.zeros(cout_pad, device=gpu)
    flush = torch.empty(256 << 20, dtype=torch.uint8, device=gpu)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    gflop = 2 * N * Ho * Wo * k * k * Cin * Cout / 1e9
    print(f"{k}x{k}/{s} {Cin} -> {Cout} @ {Ho}x{Wo} b{N}: {gflop:.1f} GFLOP, floor at 157.3 TF/s {gflop / 157.3 * 1e3:.1f} us   (act {act})")
    first = None
    for t in capi.conv2d_tactics_f32(N, H, W, Cin, Cout, k, s, p):
        if ONLY and tuple(t) != ONLY:
            continue
        if KINDS and (t[2] not in KINDS or t[3] != 16):
            continue
        y = capi.conv2d_nhwc_f32(x, wg, bias, Cout, k, k, s, p, act, tile=t)
        torch.cuda.synchronize()
        ts = []
        for _ in range(REPS):
            flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            capi.conv2d_nhwc_f32(x, wg, bias, Cout, k, k, s, p, act, out=y, tile=t)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        got = y.cpu()
        if first is None:
            first = got
        same = "same bits" if torch.equal(got, first) else f"DIFFERS max {float((got - first).abs().max()):.3g}"
        med = sorted(ts)[len(ts) // 2]
        tiles = -(-N * Ho * Wo // t[1]) * (cout_pad // t[0])
        print(f"   bn {t[0]:3d} bm {t[1]:3d} { {3: 'ptch', 5: 'regs', 6: 'role', 7: 'res3', 8: 'res1'}.get(t[2], 'dma ')} bk {t[3]:2d}  {tiles:6d} tiles ({tiles / 256:5.1f}/CU)  median {med:7.1f} us  min {min(ts):7.1f}  = {gflop / med * 1e3:6.1f} TF/s ({gflop / med * 1e3 / 157.3:.2f})  {same}")


if __name__ == "__main__":
    argv = sys.argv[1:]
    act = "silu"
    if "--act" in argv:
        i = argv.index("--act")
        act = argv[i + 1]
        del argv[i:i + 2]
    if "--only" in argv:
        i = argv.index("--only")
        ONLY = tuple(int(v) for v in argv[i + 1].split(","))
        del argv[i:i + 2]
    if "--kinds" in argv:
        i = argv.index("--kinds")
        KINDS = tuple(int(v) for v in argv[i + 1].split(","))
        del argv[i:i + 2]
    a = [int(v) for v in argv]
    shapes = [tuple(a[i:i + 7]) for i in range(0, len(a), 7)] if a else SHAPES
    for sh in shapes:
        one(*sh, act)
