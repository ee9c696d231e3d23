"""Run one conv shape N times with both kernels (for rocprofv3 --pmc)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib
dev = torch.device("cuda:0")
cin, cout, k, s, hin = [int(v) for v in sys.argv[1:6]]
p = k // 2
w = np.random.default_rng(1).normal(0, 0.05, size=(cout, cin, k, k)).astype(np.float32)
pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
wp = torch.from_numpy(pk.view(np.int16)).to(dev)
bias = torch.zeros(cp, device=dev)
N = int(os.environ.get("CONV_N", "32"))   # images (RoIs) in the batch
x = torch.randn(N, hin, hin, cin, device=dev).half()
ho = (hin + 2 * p - k) // s + 1
y = torch.empty(N, ho, ho, cout, device=dev, dtype=torch.float16)
if os.environ.get("TRTX_TACTIC"):  # "bn,bk,bm,wsk,ws,r3": pin the launch configuration (capi.conv2d_tactics lists them)
    capi.conv_force_tactic(tuple(int(v) for v in os.environ["TRTX_TACTIC"].split(",")))
for _ in range(5):
    capi.conv2d_nhwc_f16(x, wp, bias, cout, k, k, s, p, "silu", out=y)
torch.cuda.synchronize()
