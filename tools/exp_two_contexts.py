"""Experiment: two execution contexts (own arenas) alternating on two streams vs one context, YOLOv8n b32 fp16 (enqueue only)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tensorrtx_amd import engine, synth
from tensorrtx_amd import wts as wts_writer
dev = torch.device("cuda:0")
path = "/tmp/exp_yolov8n.wts"
if not os.path.exists(path):
    wts_writer.write_wts(path, synth.yolov8n_state(seed=0), dialect="double")
B, H, W = 32, 640, 640
plan = engine.build_plan("yolov8n", path, batch=B, h=H, w=W, fp16=1)
for nctx in [int(v) for v in os.environ.get("NCTX", "1,2,3").split(",")]:
    engs = [engine.Engine(plan) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    sets = []
    for e in engs:
        x = torch.from_numpy(synth.images(B, H, W, seed=1)).to(dev)
        sets.append([x if e.is_input[i] else torch.empty(B * int(np.prod(e.dims[i])), dtype=torch.float32, device=dev) for i in range(e.nb_bindings)])
    def step(k):
        j = k % nctx
        engs[j].enqueue(B, sets[j], stream=streams[j].cuda_stream)
    for k in range(12): step(k)
    torch.cuda.synchronize()
    n = 60
    t0 = time.perf_counter()
    for k in range(n): step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{nctx} context(s): {dt*1e3:.3f} ms per batch of 32, {B/dt:.0f} img/s", flush=True)
    for e in engs: e.close()
