"""Experiment: is one batch-32 enqueue slower than k concurrent batch-32/k enqueues on k streams?  (YOLOv8n 640x640, fp16)"""
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
for split in (1, 2, 4):
    b = B // split
    plan = engine.build_plan("yolov8n", path, batch=b, h=H, w=W, fp16=1)
    engs = [engine.Engine(plan) for _ in range(split)]
    streams = [torch.cuda.Stream() for _ in range(split)]
    sets = []
    for e in engs:
        x = torch.from_numpy(synth.images(b, H, W, seed=1)).to(dev)
        bind = [x if e.is_input[i] else torch.empty(b * int(np.prod(e.dims[i])), dtype=torch.float32, device=dev) for i in range(e.nb_bindings)]
        sets.append(bind)
    def step():
        for e, s, bind in zip(engs, streams, sets):
            e.enqueue(b, bind, stream=s.cuda_stream)
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"split {split} (batch {b} x {split} streams): {dt*1e3:.3f} ms per 32 images, {B/dt:.0f} img/s", flush=True)
    del engs, sets
