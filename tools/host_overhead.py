"""How long does the host spend inside one enqueue (launches + event fences) vs the GPU step?"""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
import torch
from tensorrtx_amd import engine, synth
from tensorrtx_amd import wts as wts_writer

dev = torch.device("cuda:0")
path = "/tmp/trtx_test_cache/bench_yolov8n_seed0.wts"
if not os.path.exists(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    wts_writer.write_wts(path, synth.yolov8n_state(0), dialect="double")
AUX = int(os.environ.get("AUX", "-1"))   # 0 = the throughput engine of bench.py's `value` (one stream per context), -1 = the latency engine (3 auxiliary streams)
plan = engine.build_plan("yolov8n", path, batch=32, h=640, w=640, fp16=1, **({"aux_streams": 0} if AUX == 0 else {}))
print("aux_streams", AUX)
e = engine.Engine(plan)
x = torch.from_numpy(synth.images(32, 640, 640, seed=1)).to(dev)
out = torch.empty((32, 1 + 1000 * 90), dtype=torch.float32, device=dev)
for _ in range(5):
    e.enqueue(32, [x, out])
torch.cuda.synchronize()
# host time: a single enqueue on an idle GPU returns as soon as everything is queued
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    e.enqueue(32, [x, out])
    ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
print("host time per enqueue: median %.3f ms" % (sorted(ts)[10] * 1e3))
t = time.perf_counter()
for _ in range(50):
    e.enqueue(32, [x, out])
torch.cuda.synchronize()
print("steady-state step: %.3f ms" % ((time.perf_counter() - t) / 50 * 1e3))

# the host alone: enqueue as fast as the host can, the queue never drained (what a host thread feeding several contexts has to sustain)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    e.enqueue(32, [x, out])
t_host = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print("200 enqueues: host returned after %.3f ms per enqueue; GPU done after %.3f ms per enqueue" % (t_host / 200 * 1e3, t_all / 200 * 1e3))
