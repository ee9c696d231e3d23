"""Hypothesis probe: does processing a batch of 32 as two concurrent half-batches (two contexts, two streams) beat one
batch-32 enqueue?  Prints ms per 32 images for both."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import torch
from tensorrtx_amd import engine, synth
from util import synth_wts

dev = torch.device("cuda:0")
path, _ = synth_wts("yolov8n")


def run(nctx, b):
    plan = engine.build_plan("yolov8n", path, batch=b, h=640, w=640, fp16=1)
    engs = [engine.Engine(plan) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    xs = [torch.from_numpy(synth.images(b, 640, 640, seed=i)).to(dev) for i in range(nctx)]
    outs = [torch.empty((b, 1 + 1000 * 90), dtype=torch.float32, device=dev) for _ in range(nctx)]

    def step():
        for e, s, x, o in zip(engs, streams, xs, outs):
            e.enqueue(b, [x, o], stream=s.cuda_stream)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 30
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    for e in engs:
        e.close()
    return dt * 1e3


print("1 x 32: %.3f ms" % run(1, 32))
print("2 x 16: %.3f ms" % run(2, 16))
print("4 x 8 : %.3f ms" % run(4, 8))
