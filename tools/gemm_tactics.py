"""Every tactic of conv_tactics() timed on the MFMA-bound layer shapes (plain large GEMMs): res5 of Faster R-CNN on 4 000 RoIs
(rcnn/rcnn.cpp:147-163 -> backbone.hpp:100-229: 14x14 RoI maps, 1x1 1024->512 stride 2 / 3x3 512->512 / 1x1 512->2048 / 1x1 2048->512),
ResNet-50 / RetinaFace bottleneck 1x1s at batch 32.  Prints microseconds, TFLOP/s and the fraction of the 2.5 PFLOP/s dense fp16 peak
per tactic.   python tools/gemm_tactics.py [shape indices]   (timing: hipEvents around 10 back-to-back launches, best of 3)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi  # noqa: E402

dev = torch.device("cuda:0")
PEAK = 2.5e15
# (name, N, H, W, Cin, Cout, k, stride)
SHAPES = [
    ("res5.0.conv1 1x1/2 1024->512 (1000 RoIs)", 1000, 14, 14, 1024, 512, 1, 2),
    ("res5.x.conv2 3x3 512->512 (4000 RoIs 7x7)", 4000, 7, 7, 512, 512, 3, 1),
    ("res5.x.conv3 1x1 512->2048 (4000 RoIs)", 4000, 7, 7, 512, 2048, 1, 1),
    ("res5.x.conv1 1x1 2048->512 (4000 RoIs)", 4000, 7, 7, 2048, 512, 1, 1),
    ("resnet50 b32 layer2 1x1 512->128 @28", 32, 28, 28, 512, 128, 1, 1),
    ("resnet50 b32 layer1 1x1 256->64->256 @56 (256 out)", 32, 56, 56, 64, 256, 1, 1),
    ("resnet50 b32 layer3 3x3 256->256 @14", 32, 14, 14, 256, 256, 3, 1),
    ("retinaface 1280 layer2 1x1 512->128 @160", 4, 160, 160, 512, 128, 1, 1),
    ("retinaface 1280 layer2 3x3 128->128 @160", 4, 160, 160, 128, 128, 3, 1),
]
if len(sys.argv) > 1 and sys.argv[1] != "all":
    SHAPES = [SHAPES[int(i)] for i in sys.argv[1].split(",")]
only_big = len(sys.argv) > 2 and sys.argv[2] == "big"   # time the large-GEMM tactic only (TRTX_BIG_VARIANT experiments)
out = []
for name, N, H, W, cin, cout, k, s in SHAPES:
    p = k // 2
    w = np.random.default_rng(1).normal(0, (2.0 / (cin * k * k)) ** 0.5, size=(cout, cin, k, k)).astype(np.float32)
    pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
    wp = torch.from_numpy(pk.view(np.int16)).to(dev)
    bias = torch.zeros(cp, device=dev)
    x = torch.randn(N, H, W, cin, device=dev).half()
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = torch.empty(N, ho, wo, cout, device=dev, dtype=torch.float16)
    flops = 2.0 * N * ho * wo * cout * cin * k * k
    print(f"{name}: M {N * ho * wo} N {cout} K {cin * k * k}  {flops / 1e9:.1f} GFLOP")
    rows = []
    try:
        for t in capi.conv2d_tactics(N, H, W, cin, cout, k, s, p):
            if only_big and tuple(t[:3]) != (128, 64, 256):
                continue
            if os.environ.get("GEMM_TACTIC") and ",".join(str(v) for v in t[:3]) != os.environ["GEMM_TACTIC"]:   # e.g. GEMM_TACTIC=128,64,128
                continue
            capi.conv_force_tactic(t)
            fn = lambda: capi.conv2d_nhwc_f16(x, wp, bias, cout, k, k, s, p, "relu", out=y)  # noqa: E731
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            rows.append(dict(tactic=list(t), us=round(best, 1), tflops=round(flops / best / 1e6, 1), frac=round(flops / (best * 1e-6) / PEAK, 3)))
    finally:
        capi.conv_force_tactic(None)
    rows.sort(key=lambda r: r["us"])
    for r in rows[:6]:
        print(f"   bn{r['tactic'][0]:4d} bk{r['tactic'][1]:3d} bm{r['tactic'][2]:4d} wsk{r['tactic'][3]} ws{r['tactic'][4]} r3{r['tactic'][5]}   {r['us']:9.1f} us  {r['tflops']:7.1f} TFLOP/s  {r['frac']:.3f}")
    big = [r for r in rows if r["tactic"][:3] == [128, 64, 256]]
    if big and big[0] is not rows[0]:
        r = big[0]
        print(f"   (large-GEMM tile: {r['us']} us, {r['frac']})")
    out.append(dict(shape=name, rows=rows))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_tactics.json", "w"), indent=1)
