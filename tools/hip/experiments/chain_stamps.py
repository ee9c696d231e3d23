"""Phase timestamps of the chain kernel (s_memtime per workgroup, wave 0): where does a workgroup's time go?"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tensorrtx_amd import capi  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
import chain_bench as cb  # noqa: E402  (runs its default sweep on import only if executed as a script)

dev = torch.device("cuda:0")
L = capi.lib()
N = int(os.environ.get("N", 32))
TILE = tuple(int(v) for v in os.environ.get("TILE", "0x0").split("x"))
for name in os.environ.get("SHAPES", "h80cls,h80box,b160c16,b80c32,b40c64").split(","):
    hw, cin, spec = cb.SHAPES[name]
    xs, stages = cb.make(hw, cin, spec)
    out = torch.empty((N, hw, hw, spec[-1][1]), dtype=torch.float16, device=dev)
    for dbg in [int(v) for v in os.environ.get("DBGS", "0,63").split(",")]:
        os.environ["TRTX_CHAIN_DBG"] = str(dbg)
        buf = torch.zeros((512, 16), dtype=torch.int64, device=dev)
        for _ in range(3):
            capi.conv_chain_nhwc_f16(xs[0], stages, out=out, tile=TILE)
        torch.cuda.synchronize()
        capi.check(L.trtx_op_conv_chain_set_stamps(ctypes.c_void_p(buf.data_ptr())), "stamps")
        capi.conv_chain_nhwc_f16(xs[1], stages, out=out, tile=TILE)
        torch.cuda.synchronize()
        capi.check(L.trtx_op_conv_chain_set_stamps(None), "stamps")
        st = buf.cpu().numpy().astype(np.int64)
        ns = 3 + 3 * len(stages) + 1
        st = st[:, :ns]
        st = st[(st > 0).all(axis=1)]
        d = np.diff(st, axis=1)
        labels = ["issue patch+w", "patch landed"] + [f"{w} s{s}" for s in range(len(stages)) for w in ("k-loop", "epilogue", "barrier")] + ["copy-out"]
        plan = capi.conv_chain_plan(N, hw, hw, cin, [s_["k"] for s_ in stages], [s_["cout"] for s_ in stages], [int(s_["residual"]) for s_ in stages], TILE)
        print(f"{name} plan {plan} dbg{dbg}: {len(st)} workgroups stamped; first-to-last start {int(st[:, 0].max() - st[:, 0].min())} ticks; median ticks per phase:")
        print("   ", ", ".join(f"{labels[i]}: {int(np.median(d[:, i]))}" for i in range(d.shape[1])), "| total", int(np.median(st[:, -1] - st[:, 0])),
              "| whole launch (min start -> max end)", int(st[:, -1].max() - st[:, 0].min()))
os.environ["TRTX_CHAIN_DBG"] = "0"
