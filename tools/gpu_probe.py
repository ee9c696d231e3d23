"""GPU-side probe run by gpurun: conv micro-benchmark over the YOLOv8n layer shapes at batch 32 and
plugin kernel timings.  Writes gpurun_out/probe.json.  (Development tool, not part of the product.)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
out = {"device": torch.cuda.get_device_name(0)}

# (Cin, Cout, k, s, Hin) x batch 32 — the heavy classes of SURVEY.md Appendix C.1
SHAPES = [(8, 16, 3, 2, 640), (16, 32, 3, 2, 320), (32, 32, 1, 1, 160), (16, 16, 3, 1, 160), (48, 32, 1, 1, 160),
          (32, 64, 3, 2, 160), (32, 32, 3, 1, 80), (128, 64, 1, 1, 80), (64, 128, 3, 2, 80), (64, 64, 3, 1, 40),
          (256, 128, 1, 1, 40), (128, 256, 3, 2, 40), (128, 128, 3, 1, 20), (384, 256, 1, 1, 20),
          (64, 64, 3, 1, 80), (64, 80, 3, 1, 80), (80, 80, 3, 1, 80), (80, 80, 1, 1, 80), (192, 64, 1, 1, 80),
          (128, 64, 3, 1, 40), (256, 80, 3, 1, 20)]
B = 32
rows = []
for (cin, cout, k, s, hin) in SHAPES:
    p = k // 2
    x = torch.randn(B, hin, hin, cin, device=dev).half()
    w = np.random.default_rng(0).normal(0, 0.05, size=(cout, cin, k, k)).astype(np.float32)
    packed, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
    wp = torch.from_numpy(packed.view(np.int16)).to(dev)
    bias = torch.zeros(cp, device=dev)
    ho = (hin + 2 * p - k) // s + 1
    y = torch.empty(B, ho, ho, cout, device=dev, dtype=torch.float16)
    for _ in range(3):
        capi.conv2d_nhwc_f16(x, wp, bias, cout, k, k, s, p, "silu", out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        capi.conv2d_nhwc_f16(x, wp, bias, cout, k, k, s, p, "silu", out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flop = 2.0 * B * ho * ho * cout * cin * k * k
    byts = 2.0 * (x.numel() + y.numel()) + 2.0 * packed.size
    rows.append(dict(cin=cin, cout=cout, k=k, s=s, hin=hin, bn=bn, ms=ms, tflops=flop / ms / 1e9,
                     gbps=byts / ms / 1e6))
    print(rows[-1], flush=True)
out["conv"] = rows

ins = [torch.from_numpy(a).to(dev) for a in synth.yolo_head_tensors(32, seed=0)]
dec = capi.yolo_decode(ins, 80, 640, 640, [8, 16, 32])
torch.cuda.synchronize()
for name, fn in [("decode", lambda: capi.yolo_decode(ins, 80, 640, 640, [8, 16, 32], out=dec)),
                 ("nms", lambda: capi.yolo_nms(dec))]:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    out[name + "_ms"] = e0.elapsed_time(e1) / 20
    print(name, out[name + "_ms"], flush=True)
out["decode_GBps"] = 32 * 84 * 8400 * 4 / out["decode_ms"] / 1e6
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/probe.json", "w"), indent=1)
