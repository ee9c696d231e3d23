"""Micro-benchmark of the fused chain kernel on the YOLOv8n b32 chain shapes: us per launch (HIP events over many back-to-back launches,
rotating buffers) for the chosen and forced tiles, with the kernel's ablation flags (TRTX_CHAIN_DBG), next to the same chain as
separate implicit-GEMM launches."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tensorrtx_amd import capi  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 32))
SHAPES = {
    "b160c16": (160, 16, [(3, 16, "silu", False), (3, 16, "silu", True)]),
    "b80c32": (80, 32, [(3, 32, "silu", False), (3, 32, "silu", True)]),
    "b40c64": (40, 64, [(3, 64, "silu", False), (3, 64, "silu", True)]),
    "b20c128": (20, 128, [(3, 128, "silu", False), (3, 128, "silu", True)]),
    "h80box": (80, 64, [(3, 64, "silu", False), (3, 64, "silu", False), (1, 64, "none", False)]),
    "h80cls": (80, 64, [(3, 80, "silu", False), (3, 80, "silu", False), (1, 80, "none", False)]),
    "h40box": (40, 128, [(3, 64, "silu", False), (3, 64, "silu", False), (1, 64, "none", False)]),
    "h40cls": (40, 128, [(3, 80, "silu", False), (3, 80, "silu", False), (1, 80, "none", False)]),
    "h20box": (20, 256, [(3, 64, "silu", False), (3, 64, "silu", False), (1, 64, "none", False)]),
    "h20cls": (20, 256, [(3, 80, "silu", False), (3, 80, "silu", False), (1, 80, "none", False)]),
}
which = os.environ.get("SHAPES", ",".join(SHAPES)).split(",")
tiles = [tuple(int(v) for v in t.split("x")) for t in os.environ.get("TILES", "0x0").split(",")]
dbgs = [int(v) for v in os.environ.get("DBGS", "0").split(",")]
kss = [v for v in os.environ.get("KS", "").split(",") if v != ""]   # weight modes to force: 0 resident, n = k-steps per ring slot; empty: launcher's choice
NBUF = 6


def make(hw, cin, spec):
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(N, hw, hw, cin, generator=g).half().to(dev) for _ in range(NBUF)]
    stages, c = [], cin
    for (k, cout, act, res) in spec:
        w = torch.randn(cout, c, k, k, generator=g) * (2.0 / (c * k * k)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        stages.append(dict(k=k, cout=cout, act=act, residual=res, w_f=w, b_f=b,
                           w=torch.from_numpy(capi.pack_chain_weights_f16(w.numpy()).view(np.int16)).to(dev), bias=b.to(dev)))
        c = cout
    return xs, stages


def timeit(fn, iters=40, repeat=1):
    """us per launch; repeat > 1: the C side launches `repeat` times per call (TRTX_OP_REPEAT), the host binding does not pace it"""
    os.environ["TRTX_OP_REPEAT"] = "1"
    for i in range(3):
        fn(i)
    os.environ["TRTX_OP_REPEAT"] = str(repeat)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(1, iters // repeat)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    os.environ["TRTX_OP_REPEAT"] = "1"
    return a.elapsed_time(b) / (n * repeat) * 1e3


def sweep():
    for name in which:
        hw, cin, spec = SHAPES[name]
        xs, stages = make(hw, cin, spec)
        outs = [torch.empty((N, hw, hw, spec[-1][1]), dtype=torch.float16, device=dev) for _ in range(NBUF)]
        # unfused: implicit-GEMM launches with the default tactic
        packed = []
        c = cin
        for s in stages:
            pk, cout_pad, kpad, bn = capi.pack_conv_weights_f16(s["w_f"].numpy(), cin_pad=c)
            bias = torch.zeros(cout_pad)
            bias[:s["cout"]] = s["b_f"]
            packed.append((torch.from_numpy(pk.view(np.int16)).to(dev), bias.to(dev)))
            c = s["cout"]
        mids = [[torch.empty((N, hw, hw, s["cout"]), dtype=torch.float16, device=dev) for s in stages] for _ in range(NBUF)]

        def unfused(i):
            cur = xs[i % NBUF]
            for j, s in enumerate(stages):
                cur = capi.conv2d_nhwc_f16(cur, packed[j][0], packed[j][1], s["cout"], s["k"], s["k"], 1, s["k"] // 2, s["act"],
                                        xs[i % NBUF] if s["residual"] else None, "none", out=mids[i % NBUF][j])
        t_un = timeit(unfused)
        flops = sum(2.0 * N * hw * hw * s["cout"] * s["k"] ** 2 * (cin if j == 0 else stages[j - 1]["cout"]) for j, s in enumerate(stages))
        line = f"{name:8s} N={N} unfused {t_un:7.1f} us |"
        best = None
        for tile in tiles:
            for ks in (kss or [None]):
                if ks is None:
                    os.environ.pop("TRTX_CHAIN_KS", None)
                else:
                    os.environ["TRTX_CHAIN_KS"] = ks
                plan = capi.conv_chain_plan(N, hw, hw, cin, [s["k"] for s in stages], [s["cout"] for s in stages], [int(s["residual"]) for s in stages], tile)
                if plan is None:
                    continue
                for dbg in dbgs:
                    os.environ["TRTX_CHAIN_DBG"] = str(dbg)
                    t = timeit(lambda i: capi.conv_chain_nhwc_f16(xs[i % NBUF], stages, out=outs[i % NBUF], tile=tile), iters=60, repeat=20)
                    line += f" {plan[0]}x{plan[1]} ks{plan[3]} lds{plan[2] // 1024}k{'' if dbg == 0 else ' dbg' + str(dbg)}: {t:6.1f} |"
                    if dbg == 0 and (best is None or t < best[0]):
                        best = (t, plan)
        os.environ.pop("TRTX_CHAIN_KS", None)
        if best:
            line += f" BEST {best[0]:.1f} us {best[1]} ({flops / best[0] / 1e6:.0f} TF)"
        os.environ["TRTX_CHAIN_DBG"] = "0"
        print(line, flush=True)


if __name__ == "__main__":
    sweep()
