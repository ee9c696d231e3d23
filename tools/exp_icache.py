"""Experiment: does a conv kernel run slower when the previous kernel on the stream was a DIFFERENT instantiation (cold instruction
cache)?  Run under rocprofv3 --kernel-trace; phases are separated by torch fill kernels and analysed by tools/exp_icache_parse.py."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi
dev = torch.device("cuda:0")

def mk(cin, cout, k, hin, batch):
    w = np.random.default_rng(1).normal(0, (2.0 / (cin * k * k)) ** 0.5, size=(cout, cin, k, k)).astype(np.float32)
    pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
    wp = torch.from_numpy(pk.view(np.int16)).to(dev)
    bias = torch.zeros(cp, device=dev)
    x = torch.randn(batch, hin, hin, cin, device=dev).half()
    y = torch.empty(batch, hin, hin, cout, device=dev, dtype=torch.float16)
    return lambda: capi.conv2d_nhwc_f16(x, wp, bias, cout, k, k, 1, k // 2, "silu", out=y)

A = mk(64, 64, 3, 40, 32)      # conv_igemm<4>
B = mk(128, 128, 3, 20, 32)    # conv_igemm<8>
C = mk(64, 80, 3, 40, 32)      # conv_igemm<5>
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
marker = torch.zeros(16, device=dev)
N = 100
def phase(fns):
    torch.cuda.synchronize(); marker.add_(1.0); torch.cuda.synchronize()
    for _ in range(N):
        for f in fns: f()
    torch.cuda.synchronize()
for f in (A, B, C): f()
phase([A]); phase([B]); phase([C]); phase([A, B]); phase([A, B, C]); phase([A, lambda: big.fill_(1)])
