"""Run the YOLOv8n b32 engine repeatedly on the same input (and on a permuted batch) and report bitwise differences of the decode records."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tensorrtx_amd import engine, synth
from util import synth_wts
gpu = torch.device("cuda:0")
path, _ = synth_wts("yolov8n")
mark = int(os.environ.get("MARK", "0"))
plan = engine.build_plan("yolov8n", path, batch=32, h=640, w=640, fp16=1, mark_heads=mark)
e = engine.Engine(plan)
x = torch.from_numpy(synth.images(32, 640, 640, seed=1)).to(gpu)
perm = torch.tensor([(7 * i + 3) % 32 for i in range(32)], device=gpu)
def run(xin):
    bufs = [xin if e.is_input[i] else torch.zeros(32 * int(np.prod(e.dims[i])), dtype=torch.float32, device=gpu) for i in range(e.nb_bindings)]
    e.enqueue(32, bufs)
    torch.cuda.synchronize()
    return {e.names[i]: bufs[i].cpu().numpy().reshape(32, -1) for i in range(e.nb_bindings) if not e.is_input[i]}
a = run(x); b = run(x); c = run(x[perm])
p = perm.cpu().numpy()
for k in a:
    same = np.array_equal(a[k], b[k])
    if k == "output":
        d = 0
        for j in range(32):
            n = int(c[k][j, 0]); 
            d += int(not np.array_equal(c[k][j, 1:1 + n * 90].reshape(n, 90)[:, :6], a[k][p[j], 1:1 + n * 90].reshape(n, 90)[:, :6]))
        print(k, "repeat identical:", same, " images differing under permutation:", d)
    else:
        diff = np.abs(c[k] - a[k][p]); print(k, "repeat identical:", same, " permuted max abs diff:", diff.max(), " rows differing:", int((diff.max(1) > 0).sum()))
