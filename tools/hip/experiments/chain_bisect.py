"""Which fused chain changes the YOLOv8n heads?  Builds one plan, runs it with every chain unfused (TRTX_FUSE_CHAINS=0), with all
fused, and with one chain fused at a time (TRTX_FUSE_CHAINS_MASK), and prints the head deltas against the unfused run."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tensorrtx_amd import engine, synth  # noqa: E402
from util import synth_wts  # noqa: E402

B, S = int(os.environ.get("B", 4)), int(os.environ.get("S", 640))
dev = torch.device("cuda:0")
path, _ = synth_wts("yolov8n")
os.environ["TRTX_TUNE"] = "0"
plan = engine.build_plan("yolov8n", path, batch=B, h=S, w=S, fp16=1, mark_heads=1)
x = torch.from_numpy(synth.images(B, S, S, seed=1)).to(dev)


def run(env):
    for k in ("TRTX_FUSE_CHAINS", "TRTX_FUSE_CHAINS_MASK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = engine.Engine(plan)
    bufs = [x] + [torch.full((B * int(np.prod(e.dims[i])),), float("nan"), dtype=torch.float32, device=dev) for i in range(1, e.nb_bindings)]
    e.enqueue(B, bufs)
    torch.cuda.synchronize()
    out = {n: bufs[i].cpu().numpy() for i, n in enumerate(e.names) if i > 0}
    e.close()
    return out


ref = run({"TRTX_FUSE_CHAINS": "0"})
print("unfused: decode counts", ref["output"].reshape(B, -1)[:, 0])
for label, env in [("all", {"TRTX_FUSE_CHAINS": "1"})] + [(f"chain {i}", {"TRTX_FUSE_CHAINS": "1", "TRTX_FUSE_CHAINS_MASK": hex(1 << i)}) for i in range(16)]:
    got = run(env)
    d = {n: float(np.nanmax(np.abs(got[n] - ref[n]))) if np.isfinite(got[n]).all() else float("nan") for n in got if n.startswith("head")}
    nan = {n: int((~np.isfinite(got[n])).sum()) for n in got if n.startswith("head")}
    print(label, "max |delta| per head", d, "non-finite", nan, "counts", got["output"].reshape(B, -1)[:, 0][:4])
