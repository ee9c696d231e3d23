"""Per-op hipEvent profile of any host-builder model on the GPU:  python tools/model_profile.py rcnn_r50c4 batch=1 fp16=1"""
import sys
import time

import os  # noqa: E402

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tensorrtx_amd import engine  # noqa: E402
from util import synth_wts  # noqa: E402

model = sys.argv[1]
opts = dict(kv.split("=") for kv in sys.argv[2:])
opts = {k: int(v) for k, v in opts.items()}
batch = opts.get("batch", 1)
path, _ = synth_wts(model)
plan = engine.build_plan(model, path, **opts)
e = engine.Engine(plan)
dev = torch.device("cuda:0")
bufs = []
g = torch.Generator().manual_seed(0)
for i in range(e.nb_bindings):
    n = int(np.prod(e.dims[i])) * batch
    if e.is_input[i]:
        bufs.append((torch.rand(n, generator=g) * (255 if model.startswith("rcnn") else 1)).to(dev))
    else:
        bufs.append(torch.zeros(n, device=dev))
for _ in range(3):
    e.enqueue(batch, bufs)
torch.cuda.synchronize()
t = time.time()
R = 10
for _ in range(R):
    e.enqueue(batch, bufs)
torch.cuda.synchronize()
print(f"{model} {opts}: {(time.time() - t) / R * 1e3:.3f} ms / enqueue (batch {batch}), arena {e.device_memory / 2**20:.0f} MiB")
prof = e.profile(batch, bufs)
ops = prof["ops"] if isinstance(prof, dict) else prof
tot = sum(o["ms"] for o in ops)
print(f"sum of ops {tot:.3f} ms over {len(ops)} ops")
bykind = {}
for o in ops:
    bykind[o["kind"]] = bykind.get(o["kind"], 0) + o["ms"]
print({k: round(v, 3) for k, v in sorted(bykind.items(), key=lambda kv: -kv[1])})
for o in sorted(ops, key=lambda o: -o["ms"])[:20]:
    print(f"  {o['ms']:8.3f} ms  {o['kind']:10s} {o['name']}")
