"""Timing probe of single conv shapes: python tools/ws_probe.py cin cout k hin batch [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi
cin, cout, k, hin, batch = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
p = k // 2
w = np.random.default_rng(1).normal(0, (2.0 / (cin * k * k)) ** 0.5, size=(cout, cin, k, k)).astype(np.float32)
pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
wp = torch.from_numpy(pk.view(np.int16)).to(dev)
bias = torch.zeros(cp, device=dev)
x = torch.randn(batch, hin, hin, cin, device=dev).half()
y = torch.empty(batch, hin, hin, cout, device=dev, dtype=torch.float16)
fn = lambda: capi.conv2d_nhwc_f16(x, wp, bias, cout, k, k, 1, p, "silu", out=y)
REPS = int(os.environ.get("TRTX_OP_REPS", "1"))
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('TRTX_WS_DBG','0'):>3s} nows={os.environ.get('TRTX_CONV_NOWS','-')} occ={os.environ.get('TRTX_WS_OCC','-')} {cin}->{cout} k{k} {hin}^2 b{batch}: {e0.elapsed_time(e1) / 20 / REPS * 1e3:.1f} us")
