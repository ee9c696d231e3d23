"""Per-layer table for the YOLOv8n b32 640 engine: measured time (hipEvent profile) vs algorithmic bytes / FLOP floors.
usage: python tools/layer_table.py [out.json]"""
import json, sys
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np, torch
from tensorrtx_amd import engine, synth
from util import synth_wts

B, S = 32, 640
path, _ = synth_wts("yolov8n")
plan = engine.build_plan("yolov8n", path, batch=B, h=S, w=S, fp16=1)
low = engine.describe_plan(plan, lowered=True)
eng = engine.Engine(plan)
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.images(B, S, S, seed=1)).to(dev)
out = torch.empty((B, 1 + 1000 * 90), dtype=torch.float32, device=dev)
for _ in range(3):
    eng.enqueue(B, [x, out])
torch.cuda.synchronize()
runs = [eng.profile(B, [x, out]) for _ in range(5)]
rows = []
tot = floor_tot = 0.0
for k, op in enumerate(low["ops"]):
    ms = float(np.median([(r["ops"] if isinstance(r, dict) else r)[k]["ms"] for r in runs]))
    by = op.get("bytes", 0) * B if op["kind"] != "conv" else (op["bytes"] - 0) * B
    fl = op.get("flops", 0) * B
    floor_us = max(by / 6.0e6, fl / 2.5e9)  # 6 TB/s achievable HBM, 2.5 PFLOP/s MFMA
    tot += ms * 1e3
    floor_tot += floor_us
    d = dict(i=k, kind=op["kind"], us=round(ms * 1e3, 1), floor_us=round(floor_us, 1), MB=round(by / 1e6, 1), GF=round(fl / 1e9, 2))
    if op["kind"] == "conv":
        d.update(cin=op["cin"], cout=op["cout"], k=op["k"][0], s=op["stride"][0], hin=op["hw_in"][0], res=op["residual"])
    rows.append(d)
    print(d)
print("total_us", round(tot, 1), "floor_us", round(floor_tot, 1))
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"))
