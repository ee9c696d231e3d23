"""The tolerance-meeting engine (a build WITHOUT BuilderFlag::kFP16: yolov8/include/config.h:1-3 USE_FP32) on the GPU: lowered plan summary, per-op
profile (one stream, events around every op), single-context and N-context step time, and the head tensors against the fp32 oracle.
    python tools/f32_engine_probe.py [--config yolov8n] [--batch 32] [--size 640] [--contexts 3] [--direct]   (--direct: TRTX_F32_DIRECT=1, the scalar kernel)
    python tools/f32_engine_probe.py --plan-only       (CPU: what the plan lowers to)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="yolov8n")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--contexts", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--direct", action="store_true")
    ap.add_argument("--plan-only", action="store_true")
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    if a.direct:
        os.environ["TRTX_F32_DIRECT"] = "1"
    from tensorrtx_amd import engine, synth
    from tensorrtx_amd import wts as wts_writer
    cache = os.environ.get("TRTX_TEST_CACHE", "/tmp/trtx_test_cache")
    os.makedirs(cache, exist_ok=True)
    path = os.path.join(cache, f"bench_{a.config}_seed0.wts")
    if not os.path.exists(path):
        wts_writer.write_wts(path, synth.STATE[a.config](seed=0), dialect="double")
    B, S = a.batch, a.size
    opts = dict(batch=B, h=S, w=S, fp16=0)
    t0 = time.perf_counter()
    plan = engine.build_plan(a.config, path, **opts)
    low = engine.describe_plan(plan, lowered=True)
    kinds = {}
    for op in low["ops"]:
        k = op["kind"] + ("/igemm" if op.get("igemm") else "")
        kinds[k] = kinds.get(k, 0) + 1
    print(f"plan: {len(low['ops'])} ops {kinds}, arena {low['arena_bytes'] / 1e6:.0f} MB, built in {time.perf_counter() - t0:.1f} s")
    if a.plan_only:
        for i, op in enumerate(low["ops"]):
            print(i, op["kind"], op["name"][:70], "igemm" if op.get("igemm") else "")
        return
    import torch
    dev = torch.device("cuda:0")
    e = engine.Engine(plan)
    x = synth.images(B, S, S, seed=3)
    xs = torch.from_numpy(x).to(dev)

    def outs():
        return [xs] + [torch.empty(B * int(np.prod(e.dims[i])), dtype=torch.float32, device=dev) for i in range(1, e.nb_bindings)]

    bufs = outs()
    e.enqueue(B, bufs)
    torch.cuda.synchronize()
    prof = e.profile(B, bufs)
    prof = e.profile(B, bufs)
    ops = prof["ops"] if isinstance(prof, dict) else prof
    lows = low["ops"]
    tot = sum(o["ms"] for o in ops)
    rows = []
    for i, o in enumerate(ops):
        lo = lows[i] if i < len(lows) and len(lows) == len(ops) else {}
        ms = o["kernel_ms"] if o.get("kernel_ms", 0) > 0 else o["ms"]
        fl = lo.get("flops_per_sample", lo.get("flops", 0)) * B if lo.get("igemm") else 0
        rows.append((ms, o["kind"], o["name"], fl, lo))
    conv = sum(r[0] for r in rows if r[3])
    flops = sum(r[3] for r in rows)
    print(f"profile (serialized): {tot:.3f} ms over {len(ops)} ops; MFMA convs {conv:.3f} ms = {flops / conv / 1e9 if conv else 0:.1f} TFLOP/s of 157.3; everything else {tot - conv:.3f} ms")
    for i, (ms, kind, name, fl, lo) in enumerate(rows):
        geo = f"{lo.get('cin', '')}->{lo.get('cout', '')} k{lo.get('k', [0])[0]} s{lo.get('stride', [0])[0]} @{lo.get('hw_out', [0, 0])[0]}x{lo.get('hw_out', [0, 0])[1]}" if kind == "conv" else ""
        print(f"   op {i:3d} {ms * 1e3:8.1f} us  {kind:10s} {fl / ms / 1e9 if fl else 0:6.1f} TF/s  {geo:28s} {name[:60]}")
    if hasattr(e, "tactics"):
        try:
            for t in e.tactics():
                print("   tactic", t)
        except Exception as ex:  # noqa: BLE001
            print("tactics:", ex)

    def bench(n_ctx):
        ctxs = [e] + [e.create_context() for _ in range(n_ctx - 1)]
        streams = [torch.cuda.Stream() for _ in range(n_ctx)]
        bb = [outs() for _ in range(n_ctx)]
        for k in range(2 * n_ctx):
            ctxs[k % n_ctx].enqueue(B, bb[k % n_ctx], stream=streams[k % n_ctx].cuda_stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(a.steps):
            ctxs[k % n_ctx].enqueue(B, bb[k % n_ctx], stream=streams[k % n_ctx].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / a.steps
        print(f"{n_ctx} context(s): {dt * 1e3:.3f} ms per batch of {B} = {B / dt:.0f} img/s")

    bench(1)
    if a.contexts > 1:
        bench(a.contexts)
    if a.config == "yolov8n" and not a.no_oracle:
        # logits: a second engine of the same build with the head tensors marked as outputs (that one runs the un-fused detect tail)
        from oracle import models_torch as mt
        from oracle import wts as owts
        from oracle import yolo_post as yp
        nb = min(B, 2)
        e2 = engine.Engine(engine.build_plan(a.config, path, mark_heads=1, batch=nb, h=S, w=S, fp16=0))
        x2 = torch.from_numpy(x[:nb]).to(dev)
        b2 = [x2] + [torch.empty(nb * int(np.prod(e2.dims[i])), dtype=torch.float32, device=dev) for i in range(1, e2.nb_bindings)]
        e2.enqueue(nb, b2)
        torch.cuda.synchronize()
        with torch.inference_mode():
            heads, strides = mt.yolov8_det(mt.Params(owts.load_wts(path)), torch.from_numpy(x[:nb]))
        worst = 0.0
        for i, h in enumerate(heads):
            got = b2[e2.names.index(f"head{i}")].reshape(nb, *h.shape[1:]).cpu()
            worst = max(worst, (got - h).abs().max().item())
        with torch.inference_mode():
            heads64, _ = mt.yolov8_det(mt.Params64(owts.load_wts(path)), torch.from_numpy(x[:nb]).double())
        worst64 = max((b2[e2.names.index(f"head{i}")].reshape(nb, *h.shape[1:]).cpu().double() - h).abs().max().item() for i, h in enumerate(heads64))
        o64 = max((h.double() - g).abs().max().item() for h, g in zip(heads, heads64))
        print(f"head tensors ({nb} images, logits up to {max(h.abs().max().item() for h in heads64):.1f}): max abs err {worst:.3g} vs the fp32 oracle, {worst64:.3g} vs the graph in double "
              f"(the fp32 oracle itself: {o64:.3g})")
        # boxes: the TIMED engine's decode buffer (fused detect tail) against the oracle's decode of its own heads
        dec = bufs[e.names.index("output")].reshape(B, -1)[:nb].cpu().numpy()
        ref = yp.decode_c([h.numpy() for h in heads], 80, S, S, strides)
        ious, tot = [], 0
        for b in range(nb):
            nr, ng = int(ref[b, 0]), int(dec[b, 0])
            R = ref[b, 1:1 + nr * 90].reshape(nr, 90)
            G = dec[b, 1:1 + ng * 90].reshape(ng, 90)
            for rec in R[R[:, 4] > 0.25]:
                tot += 1
                cand = G[G[:, 5] == rec[5]]
                if not len(cand):
                    ious.append(0.0)
                    continue
                ix = np.maximum(0, np.minimum(cand[:, 2], rec[2]) - np.maximum(cand[:, 0], rec[0]))
                iy = np.maximum(0, np.minimum(cand[:, 3], rec[3]) - np.maximum(cand[:, 1], rec[1]))
                inter = ix * iy
                iou = inter / ((cand[:, 2] - cand[:, 0]) * (cand[:, 3] - cand[:, 1]) + (rec[2] - rec[0]) * (rec[3] - rec[1]) - inter)
                ious.append(float(iou.max()))
        print(f"decoded boxes of the timed engine vs oracle: {tot} oracle candidates (conf > 0.25), min IoU {min(ious) if ious else None}, counts {[int(dec[b, 0]) for b in range(nb)]} vs {[int(ref[b, 0]) for b in range(nb)]}")
        e2.close()
    e.close()


if __name__ == "__main__":
    main()
