"""A/B timing of the igemm conv on the YOLOv8n layer shapes (batch 32). Usage: python tools/conv_ab.py [tag]
Honours TRTX_CONV_NOXCD / TRTX_CONV_BK32 in the environment; prints per-shape microseconds and the sum."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi

dev = torch.device("cuda:0")
SHAPES = [(16, 32, 3, 2, 320), (32, 32, 1, 1, 160), (16, 16, 3, 1, 160), (48, 32, 1, 1, 160), (32, 64, 3, 2, 160), (64, 64, 1, 1, 80),
          (32, 32, 3, 1, 80), (128, 64, 1, 1, 80), (64, 128, 3, 2, 80), (128, 128, 1, 1, 40), (64, 64, 3, 1, 40), (256, 128, 1, 1, 40),
          (128, 256, 3, 2, 40), (256, 256, 1, 1, 20), (128, 128, 3, 1, 20), (384, 256, 1, 1, 20), (512, 256, 1, 1, 20), (384, 128, 1, 1, 40),
          (192, 128, 1, 1, 40), (192, 64, 1, 1, 80), (96, 64, 1, 1, 80), (64, 64, 3, 2, 80), (128, 128, 3, 2, 40),
          (64, 64, 3, 1, 80), (64, 80, 3, 1, 80), (80, 80, 3, 1, 80), (80, 80, 1, 1, 80), (128, 64, 3, 1, 40), (128, 80, 3, 1, 40),
          (80, 80, 3, 1, 40), (256, 64, 3, 1, 20), (256, 80, 3, 1, 20), (64, 64, 3, 1, 20), (80, 80, 3, 1, 20)]
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
if len(sys.argv) > 2:  # optional subset: indices into SHAPES
    SHAPES = [SHAPES[int(i)] for i in sys.argv[2].split(",")]
rows, total = [], 0.0
for (cin, cout, k, s, hin) in SHAPES:
    p = k // 2
    w = np.random.default_rng(1).normal(0, (2.0 / (cin * k * k)) ** 0.5, size=(cout, cin, k, k)).astype(np.float32)
    pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
    wp = torch.from_numpy(pk.view(np.int16)).to(dev)
    bias = torch.zeros(cp, device=dev)
    x = torch.randn(32, hin, hin, cin, device=dev).half()
    ho = (hin + 2 * p - k) // s + 1
    y = torch.empty(32, ho, ho, cout, device=dev, dtype=torch.float16)
    fn = lambda: capi.conv2d_nhwc_f16(x, wp, bias, cout, k, k, s, p, "silu", out=y)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    total += us
    rows.append(dict(cin=cin, cout=cout, k=k, s=s, hin=hin, us=round(us, 2)))
print(tag, "sum_us", round(total, 1))
print(tag, " ".join(f"{r['us']:.1f}" for r in rows))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(f"gpurun_out/conv_ab_{tag}.json", "w"))
