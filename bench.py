#!/usr/bin/env python3
"""Headline benchmark: images/sec of the detection hot path on N MI355X GPUs of one node (BASELINE.json metric).

Default workload (config C3, BASELINE configs[2]): YOLOv8n fp16, 640x640, batch 32 per GPU: one "step" = one pass of
the hot path over one batch of synthetic images already resident in HBM = IExecutionContext::enqueue (fused MFMA
convolutions + pools/resizes + DFL tail + YoloLayer decode) followed by the GPU NMS.  K steps are K full batch-32 passes; they are
issued round-robin through --contexts execution contexts of one engine (ICudaEngine::createExecutionContext, one HIP stream each),
so batches overlap on the chip; the same K steps through a single context are timed too and printed as `single_context`.  Independent image streams are
sharded over the ranks (one process per GPU, one engine replica each, no data-path collective).

  python bench.py                                  1 GPU, C3, weak scaling unit; 3 execution contexts in flight (one stream each)
  python bench.py --contexts 1                     the reference's loop shape: one context, batches strictly in sequence
  python bench.py --gpus 8                         spawns 8 ranks itself (torch.distributed.run, RCCL barrier/all-reduce for timing)
  python -m torch.distributed.run ... bench.py --gpus 8     (how the driver launches it; same code path)
  python bench.py --gpus 8 --mode strong           global batch fixed at 32 -> 4 images per GPU
  python bench.py --config retinaface_r50 --gpus 8 C4: RetinaFace-R50 1280x1280, global batch 8, image-sharded (1 per GPU at N=8)
  python bench.py --config resnet50 | rcnn_r50c4   C2 / C5 lines (conv backbone only / full plugin chain)

Prints ONE JSON line (rank 0) with the driver contract fields plus `roofline`, `cpu_baseline` (+ `parity`, C3 at N=1).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Kernel arguments in device memory (the HIP runtime's own switch, read when it initialises - i.e. before torch is imported).  Every conv launch
# starts by reading its ~270-byte ConvArgs; from host-visible memory that first read is a ~2 us round trip over the host link: measured on one box,
# same build, HIP_FORCE_DEV_KERNARG=0 / 1: 1.113 / 1.011 ms of serialized conv time per step, 35.9k / 37.4k img/s (round 4).  ROCm 7.2's default
# already behaves like 1 on the boxes of this round; set explicitly so that a box configured otherwise measures the same thing.  A caller's own value wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

MFMA_PEAK_TFLOPS = 2500.0        # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3     # v_mfma_f32_16x16x4_f32 / 32x32x2_f32: fp32 operands, exact fp32 (MI355X_MICROARCH.md) - the fp32 engine's convolutions
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming copy reaches)
N_INPUT_SETS = 8                 # rotating input batches: a step never re-reads the batch the previous step left in the caches

# model -> (BASELINE config, per-GPU batch at weak scaling, H, W, input scale, global batch of the "strong" partition)
CONFIGS = {
    "yolov8n": dict(tag="C3 (BASELINE configs[2])", batch=32, h=640, w=640, scale=1.0, inp="images", nms=True),
    "resnet50": dict(tag="C2 (BASELINE configs[1])", batch=32, h=224, w=224, scale=1.0, inp="data", nms=False),
    "retinaface_r50": dict(tag="C4 (BASELINE configs[3])", batch=8, h=1280, w=1280, scale=255.0, inp="data", nms=False, strong_default=True),
    "rcnn_r50c4": dict(tag="C5 (BASELINE configs[4]; the reference implements R50-C4, SURVEY 8 note)", batch=4, h=800, w=1333, scale=255.0,
                       inp="images", nms=False),
}


def _traffic_from_profile(model):
    """HBM bytes per conv launch from the L2 fabric counters: measured by a SEPARATE `rocprofv3 --pmc` pass over this same
    command (tools/pmc_bench_traffic.sh) and committed under profiles/; read from there so that the number printed is the
    number on file (None when no pass has been recorded for this model)."""
    p = os.path.join(ROOT, "profiles", "pmc_conv_traffic.json")
    try:
        rec = json.load(open(p)).get(model)
        return (rec["bytes_per_launch"], rec["source"]) if rec else (None, None)
    except Exception:  # noqa: BLE001
        return None, None


def _kernel_duration_from_profile(model):
    """Mean duration of the fused MFMA conv launches as rocprofv3 --kernel-trace --stats saw them (dispatch begin -> end) in the
    committed single-context, single-lane profile of this same command: the cross-check of bench.py's live figure (which comes
    from HIP events attached to each launch - the same begin / end timestamps - or, failing that, from the stream events around
    each op, which also contain the hand-over between two launches)."""
    import re
    import glob
    cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_c3_1ctx_lanes1.txt"))) if model == "yolov8n" else []
    if not cand:
        return None, None
    name = os.path.basename(cand[-1])   # the latest round's
    try:
        calls, tot = 0, 0.0
        for line in open(os.path.join(ROOT, "profiles", name)):
            if "conv_igemm" in line or "conv_ws" in line or "conv_gemm256" in line or "conv_patch" in line or "conv_res" in line:
                m = re.search(r"\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
                if m:
                    calls += int(m.group(1))
                    tot += float(m.group(5))
        return (tot * 1e3 / calls, "profiles/" + name) if calls else (None, None)
    except Exception:  # noqa: BLE001
        return None, None


def cpu_baseline_and_parity(path, gpu_heads, gpu_dec, images, seconds_budget=20.0, precision="fp16", extra=None):
    """The oracle (PyTorch-CPU fp32 restatement of the reference graph + C decode/NMS) timed on the host cores on a bounded
    sample of the same workload: reported next to the GPU number, never the thing measured.  The same oracle outputs are
    compared with what the GPU produced for the same images -> `parity` (what the fp16 engine is off by)."""
    import numpy as np
    import torch

    from oracle import models_torch as mt
    from oracle import wts as owts
    from oracle import yolo_post as yp

    cores = min(os.cpu_count() or 1, 32)  # small-channel convolutions stop scaling (and oversubscribe) past a few dozen threads
    torch.set_num_threads(cores)
    params = owts.load_wts(path)
    nb = 4
    x = torch.from_numpy(images[:nb])
    keep = {}

    def once():
        with torch.inference_mode():
            heads, strides = mt.yolov8_det(mt.Params(params), x)
        dec = yp.decode_c([h.numpy() for h in heads], 80, x.shape[2], x.shape[3], strides)
        keep["heads"], keep["dec"] = heads, dec
        keep["nms"] = yp.batch_nms_c(dec)

    with torch.inference_mode():   # the same graph in double (untimed): the value every fp32 evaluation - the oracle's own included - is a rounding of
        keep["heads64"] = mt.yolov8_det(mt.Params64(params), x.double())[0]

    once()  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        once()
        n += nb
        if time.perf_counter() - t0 > seconds_budget or n >= 64:
            break
    dt = time.perf_counter() - t0
    base = {"value": n / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} images ({nb}/iter) 640x640 fp32, PyTorch-CPU restatement of the reference graph + C decode/NMS"}
    # parity of the fp16 engine vs the fp32 oracle on those same images: head logits and decoded boxes
    if extra:   # further engines against the same oracle outputs: {name: (heads, decode buffer, description)} -> parity objects under "extra"
        more = {}
        for name, (eh, ed, what) in extra.items():
            more[name] = _yolo_parity(keep, eh, ed, nb, what)
    par = _yolo_parity(keep, gpu_heads, gpu_dec, nb, "fp16 build (fp16 storage of 63 layers)" if precision == "fp16" else
                       "int8 build (int8 activations and weights behind the first two layers, fp16 elsewhere)")
    if extra:
        par["extra"] = more
    return base, par


def _yolo_parity(keep, gpu_heads, gpu_dec, nb, what):
    import numpy as np
    logit_err = max(float((g[:nb] - h.numpy().reshape(g[:nb].shape)).__abs__().max()) for g, h in zip(gpu_heads, keep["heads"]))
    logit_max = max(float(h.abs().max()) for h in keep["heads"])
    err64 = max(float(np.abs(g[:nb].astype(np.float64) - h.numpy().reshape(g[:nb].shape)).max()) for g, h in zip(gpu_heads, keep["heads64"]))
    oracle64 = max(float((h.double() - h64).abs().max()) for h, h64 in zip(keep["heads"], keep["heads64"]))
    ious, matched, total, loose = [], 0, 0, []
    for b in range(nb):
        nr, ng = int(keep["dec"][b, 0]), int(gpu_dec[b, 0])
        r = keep["dec"][b, 1:1 + nr * 90].reshape(nr, 90)
        g = gpu_dec[b, 1:1 + ng * 90].reshape(ng, 90)
        for rec in r[r[:, 4] > 0.25]:
            total += 1
            cand = g[g[:, 5] == rec[5]]
            if not len(cand):
                continue
            ix = np.maximum(0, np.minimum(cand[:, 2], rec[2]) - np.maximum(cand[:, 0], rec[0]))
            iy = np.maximum(0, np.minimum(cand[:, 3], rec[3]) - np.maximum(cand[:, 1], rec[1]))
            inter = ix * iy
            iou = inter / ((cand[:, 2] - cand[:, 0]) * (cand[:, 3] - cand[:, 1]) + (rec[2] - rec[0]) * (rec[3] - rec[1]) - inter)
            if iou.max() > 0.5:
                loose.append(float(iou.max()))
            if iou.max() > 0.9:
                matched += 1
                ious.append(float(iou.max()))
    met = logit_err <= 1e-4 and bool(ious) and min(ious) >= 0.999 and matched == total
    parity = {"vs": "fp32 PyTorch-CPU oracle, same weights and images", "images": nb, "head_logit_max_abs_err": logit_err, "largest_oracle_logit": logit_max,
              "head_logit_max_abs_err_vs_the_graph_in_double": err64, "the_fp32_oracle_vs_the_graph_in_double": oracle64,
              "oracle_candidates_conf>0.25": total, "matched_same_class_iou>0.9": matched, "matched_same_class_iou>0.5": len(loose),
              "min_box_iou": min(ious) if ious else None, "mean_box_iou_of_iou>0.5_matches": float(np.mean(loose)) if loose else None,
              "north_star_tolerance": ("1e-4 logit / 1e-3 IoU: MET by this " if met else "1e-4 logit / 1e-3 IoU: NOT met by this ") + what,
              "nms_kept_indices": "bit-exact vs the oracle on identical decode buffers (tests/test_gpu_yolo_plugins.py, test_ref_pinning.py)"}
    return parity


def _box_match(R, G, iou_floor=0.5):
    """for every reference box (rows of R: x1 y1 x2 y2) the best IoU among the boxes of G -> (best IoU per reference box)"""
    import numpy as np
    if not len(R) or not len(G):
        return np.zeros(len(R))
    best = np.zeros(len(R))
    for s in range(0, len(R), 512):
        r = R[s:s + 512, None, :]
        ix = np.maximum(0, np.minimum(r[..., 2], G[None, :, 2]) - np.maximum(r[..., 0], G[None, :, 0]))
        iy = np.maximum(0, np.minimum(r[..., 3], G[None, :, 3]) - np.maximum(r[..., 1], G[None, :, 1]))
        inter = ix * iy
        ua = (r[..., 2] - r[..., 0]) * (r[..., 3] - r[..., 1]) + ((G[:, 2] - G[:, 0]) * (G[:, 3] - G[:, 1]))[None, :] - inter
        best[s:s + 512] = (inter / np.maximum(ua, 1e-12)).max(1)
    return best


def cpu_baseline_and_parity_other(config, path, H, W, sample, gpu_out, seconds_budget=20.0):
    """cpu_baseline + parity for C2 / C4 / C5: the oracle's PyTorch-CPU fp32 twin of the reference graph (+ the NumPy /
    C restatement of its plugins) timed on a bounded sample of the same workload, and compared with what the TIMED engine returned for those
    images.  `sample`: the network input of the first images of one of the timed batches (numpy, network layout); gpu_out: name -> numpy."""
    import numpy as np
    import torch

    from oracle import det_post as dp
    from oracle import models_torch as mt
    from oracle import wts as owts

    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    params = owts.load_wts(path)
    x = torch.from_numpy(sample)
    nb = x.shape[0]
    keep = {}

    def once():
        with torch.inference_mode():
            if config == "resnet50":
                keep["logits"] = mt.resnet50(mt.Params(params), x).numpy()
            elif config == "retinaface_r50":
                heads = mt.retinaface_r50(mt.Params(params), x)
                keep["dec"] = dp.retina_decode([h.reshape(nb, 32, -1).numpy() for h in heads], H, W)
            else:
                keep["det"] = mt.rcnn_r50c4(mt.Params(params), x)

    t0 = time.perf_counter()
    once()
    n = nb
    first = time.perf_counter() - t0
    while time.perf_counter() - t0 + first < seconds_budget and n < 64:   # (no separate warm-up: one R-CNN image is already ~10 s of CPU)
        once()
        n += nb
    dt = time.perf_counter() - t0
    what = {"resnet50": "224x224 fp32, PyTorch-CPU restatement of resnet/resnet50.cpp",
            "retinaface_r50": f"{H}x{W} fp32, PyTorch-CPU restatement of retinaface/retina_r50.cpp + NumPy Decode_TRT",
            "rcnn_r50c4": f"{H}x{W} fp32, PyTorch-CPU restatement of rcnn/rcnn.cpp + NumPy restatements of its six plugins"}[config]
    base = {"value": n / dt, "unit": "images/sec", "cores": cores, "kind": "port", "sample": f"{n} images ({nb}/iter) {what}"}
    par = {"vs": "fp32 PyTorch-CPU oracle, same weights and images", "images": nb, "engine": "the TIMED engine (its own batch, fp16)"}
    if config == "resnet50":
        g = gpu_out["prob"].reshape(-1, keep["logits"].shape[1])[:nb]
        err = float(np.abs(g - keep["logits"]).max())
        par.update(logit_max_abs_err=err, logit_max=float(np.abs(keep["logits"]).max()), rel_to_largest_logit=err / float(np.abs(keep["logits"]).max()),
                   top1_agree=float((g.argmax(1) == keep["logits"].argmax(1)).mean()))
    elif config == "retinaface_r50":
        ref, dec = keep["dec"], gpu_out["prob"].reshape(-1, keep["dec"].shape[1])[:nb]
        tot = hit = 0
        ious = []
        for b in range(nb):
            nr, ng = int(ref[b, 0]), int(dec[b, 0])
            R = ref[b, 1:1 + nr * 15].reshape(nr, 15)
            G = dec[b, 1:1 + ng * 15].reshape(ng, 15)
            R = R[R[:, 4] > 0.1]
            best = _box_match(R[:, :4], G[:, :4])
            tot += len(R)
            hit += int((best > 0.9).sum())
            ious += best[best > 0.9].tolist()
        par.update({"oracle_candidates_conf>0.1": tot, "matched_iou>0.9": hit, "min_box_iou_of_matches": min(ious) if ious else None,
                    "decode_counts": [int(dec[b, 0]) for b in range(nb)], "oracle_counts": [int(ref[b, 0]) for b in range(nb)]})
    else:
        det = keep["det"]
        D = det["boxes"].shape[1]
        gb = gpu_out["boxes"].reshape(-1, D, 4)[:nb]
        gs = gpu_out["scores"].reshape(-1, D)[:nb]
        gl = gpu_out["labels"].reshape(-1, D)[:nb]
        tot = hit = 0
        for b in range(nb):
            valid = det["scores"][b] > 0.05
            R = np.asarray(det["boxes"][b])[valid]
            best = _box_match(R, gb[b][gs[b] > 0.0])
            tot += len(R)
            hit += int((best > 0.5).sum())
        par.update({"oracle_detections_score>0.05": tot, "matched_iou>0.5": hit, "top_score_err": float(np.abs(gs[:, 0] - np.asarray(det["scores"])[:, 0]).max()),
                    "labels_equal_fraction": float((gl == np.asarray(det["labels"])).mean()),
                    "note": "end to end: one flipped top-k / NMS decision upstream legitimately changes what follows; stage-wise parity in tests/test_gpu_rcnn.py"})
    par["north_star_tolerance"] = "1e-4 logit / 1e-3 IoU: met by the fp32 builds (tests), not by fp16 storage (tests/parity.py CEILINGS states the fp16 budget)"
    return base, par


def _spawn_self(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run and relay the JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def run_in_process(args):
    """--replicas in-process: ONE process drives --gpus N devices through replicas.DeviceReplicas - the reference's own recipe for one
    host with several GPUs (tutorials/multi_GPU_processing.md:13-30: cudaSetDevice(i), one {engine, context, stream, buffers} per device).
    Same plan on every device, --contexts execution contexts per device, images resident per device, no cross-device traffic; a "step"
    enqueues one batch on EVERY device.  The driver's contract is one process per GPU (the default path); this is the alternative for
    callers that own all GPUs from one address space.  YOLOv8n (C3) only."""
    import numpy as np
    import torch

    dry = args.dry_run
    n_dev = args.gpus
    if dry:
        from tensorrtx_amd import dryrun
        tc = dryrun.FakeCuda(n_dev)
    else:
        tc = torch.cuda
        if not tc.is_available() or tc.device_count() < n_dev:
            raise SystemExit(f"--replicas in-process --gpus {n_dev}: {tc.device_count() if tc.is_available() else 0} GPU(s) visible")
    from tensorrtx_amd import capi, engine, replicas, synth
    from tensorrtx_amd import wts as wts_writer

    if args.config != "yolov8n" or args.precision != "fp16":
        raise SystemExit("--replicas in-process: yolov8n fp16 only")
    cfg = CONFIGS["yolov8n"]
    batch, H, W = cfg["batch"], cfg["h"], cfg["w"]
    cache = os.environ.get("TRTX_TEST_CACHE", "/tmp/trtx_test_cache")
    os.makedirs(cache, exist_ok=True)
    path = os.path.join(cache, "bench_yolov8n_seed0.wts")
    if not os.path.exists(path):
        tmp = path + f".{os.getpid()}.tmp"
        wts_writer.write_wts(tmp, synth.yolov8n_state(seed=0), dialect="double")
        os.replace(tmp, path)
    n_ctx = max(1, args.contexts)
    if not dry:
        tc.set_device(0)
    plan = engine.build_plan("yolov8n", path, batch=batch, h=H, w=W, fp16=1, aux_streams=0 if n_ctx > 1 else -1)   # built (tactics timed) next to device 0
    low = engine.describe_plan(plan, lowered=True)
    make_engine = (lambda d: dryrun.DeviceEngine(plan, engine.describe_plan)) if dry else (lambda d: engine.Engine(plan))
    if dry:
        reps = dryrun.Replicas(range(n_dev), make_engine)
    else:
        reps = replicas.DeviceReplicas(range(n_dev), make_engine)
    L = None if dry else capi.lib()
    if L:
        L.trtx_yolo_nms_workspace.restype = ctypes.c_size_t
    imgs = synth.images(batch, H, W, seed=100)

    class Slot:
        def __init__(self, d, e, ctx):
            dev = torch.device("cpu") if dry else torch.device("cuda", d)
            self.d, self.ctx = d, ctx
            self.stream = tc.Stream() if dry else tc.Stream(device=d)
            self.x = torch.from_numpy(imgs).to(dev)
            self.out = (torch.zeros if dry else torch.empty)((batch, 1 + 1000 * 90), dtype=torch.float32, device=dev)
            self.keep_idx = torch.empty((batch, 1000), dtype=torch.int32, device=dev)
            self.keep_cnt = torch.zeros((batch,), dtype=torch.int32, device=dev)
            self.keep_det = torch.empty((batch, 1000, 6), dtype=torch.float32, device=dev)
            self.ws_bytes = L.trtx_yolo_nms_workspace(batch) if L else 256
            self.ws = torch.empty((self.ws_bytes,), dtype=torch.uint8, device=dev)

        def run(self):
            self.ctx.enqueue(batch, [self.x, self.out], stream=self.stream.cuda_stream)
            if L:
                capi.check(L.trtx_yolo_nms(capi._p(self.out), batch, 1000, ctypes.c_float(0.5), ctypes.c_float(0.45), capi._p(self.keep_idx), capi._p(self.keep_cnt),
                                           capi._p(self.keep_det), capi._p(self.ws), ctypes.c_size_t(self.ws_bytes), ctypes.c_void_p(self.stream.cuda_stream)), "trtx_yolo_nms")

    slots = []   # slots[d][j]
    for d, e in zip(reps.devices, reps.engines):
        if not dry:
            tc.set_device(d)
        slots.append([Slot(d, e, e if j == 0 else e.create_context()) for j in range(n_ctx)])

    def sync_all():
        for d in reps.devices:
            if not dry:
                tc.set_device(d)
            tc.synchronize()

    def step(k):
        for d in range(n_dev):     # engines are bound to their device: make it current for the launches
            if not dry:
                tc.set_device(reps.devices[d])
            slots[d][k % n_ctx].run()

    for k in range(max(24, 8 * n_ctx)):   # settle
        step(k)
    sync_all()
    legs = []
    for _ in range(max(1, args.repeats)):
        for k in range(args.warmup):
            step(k)
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        sync_all()
        legs.append(time.perf_counter() - t0)
    dt = sorted(legs)[len(legs) // 2]
    if not dry:
        tc.set_device(reps.devices[0])
    kept = float(slots[0][0].keep_cnt.float().mean().item())
    res = {
        "metric": f"images/sec @ batch={batch} {W}x{H} fp16 (yolov8n conv backbone + YoloLayer decode + NMS)", "value": n_dev * batch * args.steps / dt, "unit": "images/sec",
        "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "legs_ms": [round(x / args.steps * 1e3, 4) for x in legs], "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"yolov8n fp16 {W}x{H}, C3 (BASELINE configs[2]): per-GPU batch {batch}, enqueue + GPU NMS, {n_ctx} execution contexts per device",
                   "contexts": n_ctx, "global_batch": n_dev * batch,
                   "parallelism": f"ONE process driving {n_dev} device(s) through DeviceReplicas (tutorials/multi_GPU_processing.md:13-30); image-sharded, no cross-device traffic"},
        "roofline": None, "cpu_baseline": None,
        "note": "in-process replicas is a secondary mode: roofline / cpu_baseline / parity are printed by the default one-process-per-GPU path",
        "detections": {"kept_after_nms_per_image": kept},
    }
    if dry:
        res["dry_run"] = True
        res["data"] = "NONE: --dry-run rehearses the control flow on CPU; every number in this line is meaningless"
    print(json.dumps(res), flush=True)
    reps.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="yolov8n", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default=None, choices=["weak", "strong"],
                    help="weak: fixed per-GPU batch (default); strong: the config's global batch is split over the GPUs")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "int8"],
                    help="int8: BuilderFlag::kINT8 engine (entropy calibration on 2 synthetic batches, int8 MFMA convs, fp16 fallback)")
    ap.add_argument("--contexts", type=int, default=3,
                    help="execution contexts kept in flight per GPU (each on its own stream; 1 = the reference's serial loop)")
    ap.add_argument("--copy-priority", type=int, default=1, help="host-fed leg: 1 = the upload stream is a high-priority stream (default), 0 = a plain one")
    ap.add_argument("--upload-buffers", type=int, default=0,
                    help="host-fed leg: device frame buffers the uploads rotate through (0 = 2 x contexts + 2)")
    ap.add_argument("--repeats", type=int, default=7,
                    help="the W-warm-up + K-step timed leg is run this many times back to back; `value` is the MEDIAN leg (all legs are printed)")
    ap.add_argument("--replicas", default="processes", choices=["processes", "in-process"],
                    help="processes (default, the driver's contract): one rank per GPU; in-process: this one process drives --gpus devices through DeviceReplicas")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pin", action="store_true", help="leave the process's CPU affinity alone (default: the cores of the GPU's NUMA node)")
    ap.add_argument("--no-tolerance-engine", action="store_true", help="skip the fp32-engine leg (yolov8n fp16 runs print it as `tolerance_engine`)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of this script's control flow (ranks over gloo, rank 0 builds and broadcasts the plan, legs / barriers / "
                         "max-over-ranks, JSON assembly) with stand-ins for every GPU object (tensorrtx_amd/dryrun.py); the line is marked dry_run and "
                         "its numbers mean nothing")
    ap.add_argument("--dump-ops", default="", help="write the per-op hipEvent timing table (JSON) to this path")
    args = ap.parse_args()

    if args.replicas == "in-process":
        return run_in_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_self(args)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    if dry:
        from tensorrtx_amd import dryrun
        tc = dryrun.FakeCuda(world)          # stands in for torch.cuda below
    else:
        tc = torch.cuda
    if not tc.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if tc.device_count() < (local + 1):
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {tc.device_count()} GPU(s) visible")
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        tc.set_device(local)
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        tc.set_device(0)
    dev = torch.device("cpu") if dry else torch.device("cuda", tc.current_device())
    pin = (lambda t: t) if dry else (lambda t: t.pin_memory())

    from tensorrtx_amd import capi, engine, replicas, synth
    # each rank on the cores of its GPU's NUMA node, before any worker thread or pinned buffer exists
    affinity_note = "dry run" if dry else ("off (--no-pin)" if args.no_pin else replicas.pin_to_gpu_numa_node(tc.current_device(), local, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    from tensorrtx_amd import wts as wts_writer

    cfg = CONFIGS[args.config]
    mode = args.mode or ("strong" if cfg.get("strong_default") else "weak")
    if mode == "strong":
        mine = replicas.partition(cfg["batch"], world, rank)
        batch = len(mine)
        if batch == 0:
            raise SystemExit(f"strong scaling of global batch {cfg['batch']} over {world} GPUs leaves rank {rank} empty")
        global_batch = cfg["batch"]
    else:
        batch = cfg["batch"]
        global_batch = batch * world
    H, W = cfg["h"], cfg["w"]

    cache = os.environ.get("TRTX_TEST_CACHE", "/tmp/trtx_test_cache")
    os.makedirs(cache, exist_ok=True)
    # seeded synthetic weights (no trained weights offline) from the PRODUCT-side generators (tensorrtx_amd/synth.py; the same draws as the test
    # suite's oracle-driven generator - tests/test_runtime_cpu.py), written once per box in the reference's .wts format
    path = os.path.join(cache, f"bench_{args.config}_seed0.wts")
    if not os.path.exists(path):
        tmp = f"{path}.{os.getpid()}.tmp"   # ranks of a multi-GPU bench may write the same file concurrently
        wts_writer.write_wts(tmp, synth.STATE[args.config](seed=0), dialect="double")
        os.replace(tmp, path)
    n_ctx = max(1, args.contexts)
    calib_cache = {}

    def build(aux):
        opts = dict(batch=batch, h=H, w=W, fp16=1)
        if aux >= 0:
            opts["aux_streams"] = aux        # IBuilderConfig::setMaxAuxStreams
        if args.precision == "int8":
            from tensorrtx_amd import calibrator
            cal_batches = [torch.from_numpy(synth.images(batch, H, W, seed=900 + k) * cfg["scale"]).to(dev) for k in range(2)]
            cal = calibrator.Calibrator(batches=cal_batches, batch_size=batch, cache=calib_cache.get("text"))
            with cal.installed():
                plan8 = engine.build_plan(args.config, path, int8=1, **opts)
            calib_cache.setdefault("text", cal.written_cache)   # later builds (other aux-stream settings, the parity engine) reuse the scales
            return plan8
        return engine.build_plan(args.config, path, **opts)

    # Several execution contexts in flight (each on its own stream, over one set of weights) is how a throughput-oriented caller
    # drives an engine; with them the concurrency comes from whole batches overlapping, so the contexts themselves stay on their
    # caller's stream (setMaxAuxStreams(0)).  A single context instead spreads independent branches over 3 auxiliary streams.
    def build_shared(aux):
        """rank 0 builds (and, next to its GPU, times the kernel tactics into the plan); every rank runs THAT plan: all replicas launch the
        same kernels and return the same bits (the optional splitter's one-off broadcast; not on the data path)"""
        plan = build(aux) if rank == 0 else None
        if dist:
            box = [plan]
            dist.broadcast_object_list(box, src=0)
            plan = box[0]
        return plan

    plan = build_shared(0 if n_ctx > 1 else -1)
    low = engine.describe_plan(plan, lowered=True)
    make_engine = (lambda pl: dryrun.DryEngine(pl, engine.describe_plan)) if dry else engine.Engine
    eng = make_engine(plan)

    # bindings: N_INPUT_SETS rotating input batches (resident in HBM), one set of outputs per context
    n_sets = N_INPUT_SETS if args.config == "yolov8n" else 2
    if dry:   # one synthetic batch stands for all of them (generating 8 x 32 images is most of a dry run's time)
        rng_imgs = [synth.images(batch, H, W, seed=100 + 17 * rank) * cfg["scale"]] * n_sets
    else:
        rng_imgs = [synth.images(batch, H, W, seed=100 + 17 * rank + k) * cfg["scale"] for k in range(n_sets)]
    nhwc_input = args.config == "rcnn_r50c4"  # DataPreprocess takes HWC images (rcnn.cpp:80-100)
    inputs = [torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1)) if nhwc_input else x).to(dev) for x in rng_imgs]
    in_idx = [i for i in range(eng.nb_bindings) if eng.is_input[i]][0]
    L = None if dry else capi.lib()
    if L:
        L.trtx_yolo_nms_workspace.restype = ctypes.c_size_t

    class Slot:
        """One batch in flight: an execution context, its stream, its output / NMS / pinned host buffers."""

        def __init__(self, e, ctx):
            self.e, self.ctx = e, ctx
            self.stream = tc.Stream()
            self.stream_p = ctypes.c_void_p(self.stream.cuda_stream)
            self.outs = {i: (torch.zeros if dry else torch.empty)(batch * int(np.prod(e.dims[i])), dtype=torch.float32, device=dev)
                         for i in range(e.nb_bindings) if not e.is_input[i]}
            if cfg["nms"]:
                self.out = self.outs[e.names.index("output")].reshape(batch, 1 + 1000 * 90)
                self.keep_idx = torch.empty((batch, 1000), dtype=torch.int32, device=dev)
                self.keep_cnt = torch.empty((batch,), dtype=torch.int32, device=dev)
                self.keep_det = torch.empty((batch, 1000, 6), dtype=torch.float32, device=dev)
                self.ws_bytes = L.trtx_yolo_nms_workspace(batch) if L else 256
                self.ws = torch.empty((self.ws_bytes,), dtype=torch.uint8, device=dev)
                self.host_cnt = pin(torch.empty((batch,), dtype=torch.int32))
                self.host_det = pin(torch.empty((batch, 1000, 6), dtype=torch.float32))

        def bindings(self, x):
            return [x if i == in_idx else self.outs[i] for i in range(self.e.nb_bindings)]

        def run(self, x, with_d2h=False, frames=None):
            """enqueue (+ device NMS (+ D2H of the detections)) of one batch, all on this slot's stream; frames: uint8 HWC camera
            frames instead of a network-input tensor (the stem samples them: trtx_context_enqueue_frames)"""
            if frames is not None and not dry:
                self.ctx.enqueue_frames(batch, frames, self.bindings(None), stream=self.stream.cuda_stream)
            else:
                self.ctx.enqueue(batch, self.bindings(x), stream=self.stream.cuda_stream)
            if cfg["nms"] and dry:
                self.keep_cnt.zero_()
            elif cfg["nms"]:
                capi.check(L.trtx_yolo_nms(capi._p(self.out), batch, 1000, ctypes.c_float(0.5), ctypes.c_float(0.45), capi._p(self.keep_idx),
                                           capi._p(self.keep_cnt), capi._p(self.keep_det), capi._p(self.ws), ctypes.c_size_t(self.ws_bytes),
                                           self.stream_p), "trtx_yolo_nms")
                if with_d2h:  # what the reference's timer includes: results back on the host (yolov8_det.cpp:97-104)
                    with tc.stream(self.stream):
                        self.host_cnt.copy_(self.keep_cnt, non_blocking=True)
                        self.host_det.copy_(self.keep_det, non_blocking=True)  # contiguous 768 KB

    def make_slots(e, n):
        return [Slot(e, e if j == 0 else e.create_context()) for j in range(n)]

    def timed(slots, n_steps, with_d2h=False, trace=None):
        """one timed leg: barrier + synchronize, EXACTLY n_steps steps, synchronize + barrier; max over ranks (seconds).
        trace (a dict): per-step host time at which the enqueue returned and device time at which the step's last kernel ended,
        both in ms from the start of the leg - what lets a reader see WHERE a slow leg lost its time."""
        tc.synchronize()
        if dist:
            dist.barrier()
        tc.synchronize()
        evs = None
        if trace is not None:
            evs = [tc.Event(enable_timing=True) for _ in range(n_steps + 1)]
            evs[0].record(slots[0].stream)
            host = []
        t0 = time.perf_counter()
        for k in range(n_steps):
            sl = slots[k % len(slots)]
            sl.run(inputs[k % len(inputs)], with_d2h)   # step k; steps on the same slot are ordered by its stream
            if evs:
                evs[k + 1].record(sl.stream)
                host.append((time.perf_counter() - t0) * 1e3)
        tc.synchronize()                     # all streams: every step's NMS / copies end inside the timed region
        if dist:
            dist.barrier()
        tc.synchronize()
        dt = time.perf_counter() - t0
        if evs:
            trace["host_enqueue_returned_ms"] = [round(h, 3) for h in host]
            trace["device_step_done_ms"] = [round(evs[0].elapsed_time(evs[k + 1]), 3) for k in range(n_steps)]
        if trace is not None:
            trace["per_rank_ms_per_step"] = [round(v / n_steps * 1e3, 4) for v in replicas.gather_over_ranks(dt, dist, dev)]   # (every rank's own clock)
        return replicas.max_over_ranks(dt, dist, dev)

    def legs(slots, repeats, with_d2h=False, warmup=None):
        """`repeats` x (W untimed warm-up steps, then one timed leg of exactly K steps).  Returns (median seconds, all legs in ms
        per step, trace of the slowest leg).  One leg is 20-50 ms of GPU work: a single sample can land on a clock ramp, a
        neighbour's rocm-smi query or a host hiccup, so the figure reported is the median and every leg is printed beside it."""
        w = args.warmup if warmup is None else warmup
        out, traces = [], []
        for _ in range(max(1, repeats)):
            for k in range(w):
                slots[k % len(slots)].run(inputs[k % len(inputs)], with_d2h)
            tr = {}
            out.append(timed(slots, args.steps, with_d2h, tr))
            traces.append(tr)
        med = sorted(out)[len(out) // 2]
        worst = max(range(len(out)), key=lambda i: out[i])
        med_i = min(range(len(out)), key=lambda i: abs(out[i] - med))
        return med, [round(o / args.steps * 1e3, 4) for o in out], dict(traces[worst], leg=worst, per_rank_ms_per_step_median_leg=traces[med_i].get("per_rank_ms_per_step"))

    slots = make_slots(eng, n_ctx)
    # the single-context engine is built (and its tactics timed) BEFORE any timed leg, so that no timed leg is the first GPU work
    # after seconds of host-only set-up
    eng1 = one = None
    if n_ctx > 1:
        eng1 = make_engine(build_shared(-1))
        one = make_slots(eng1, 1)
    # settle (set-up, untimed, before the W warm-up steps the contract asks for): every slot, stream and input batch has been used
    # and the clocks are up: at least 24 steps AND at least 0.4 s of back-to-back work
    t_settle = time.perf_counter()
    k = 0
    while k < max(24, 8 * n_ctx) or time.perf_counter() - t_settle < 0.4:
        slots[k % n_ctx].run(inputs[k % len(inputs)])
        k += 1
        if k % 8 == 0:
            tc.synchronize()
    tc.synchronize()
    settle_steps = k
    dt, value_legs, value_trace = legs(slots, args.repeats)
    affinity_all = replicas.gather_notes(affinity_note, dist)   # every rank's pinning report (a mis-pinned straggler would otherwise be invisible)
    detections = None
    if cfg["nms"]:
        tc.synchronize()
        last = slots[(args.steps - 1) % n_ctx]
        detections = {"decode_candidates_per_image": float(last.out[:, 0].float().mean().item()),
                      "kept_after_nms_per_image": float(last.keep_cnt.float().mean().item()),
                      "note": "last timed step; seeded random weights: counts are not those of a trained model"}
    n_side = max(3, min(args.repeats, 5))
    dt_d2h, d2h_legs, _ = legs(slots, n_side, with_d2h=True) if cfg["nms"] else (None, None, None)
    # the reference's own loop shape for comparison: ONE context, batches strictly one after the other (3 auxiliary streams inside it)
    dt_single = single_legs = None
    if n_ctx > 1:
        dt_single, single_legs, _ = legs(one, n_side, warmup=min(args.warmup, 5))

    # The tolerance-meeting engine: the SAME network built WITHOUT BuilderFlag::kFP16 - the reference's USE_FP32 build
    # (yolov8/include/config.h:1-3, model.cpp:314-324) - whose convolutions run on the fp32 MFMA (kernels/conv_igemm_f32.hip).  BASELINE's
    # tolerance (1e-4 logit / 1e-3 IoU against the reference's fp32 outputs) is a property of fp32 arithmetic, so this is the engine it can be
    # stated for; timed exactly like `value` (same contexts in flight, same inputs, barrier + synchronize, max over ranks), never `value`.
    tol = None
    if args.config == "yolov8n" and args.precision == "fp16" and not args.no_tolerance_engine:
        def build32(aux):
            plan32 = None
            if rank == 0:
                opts = dict(batch=batch, h=H, w=W, fp16=0)
                if aux >= 0:
                    opts["aux_streams"] = aux
                plan32 = engine.build_plan(args.config, path, **opts)
            if dist:
                box = [plan32]
                dist.broadcast_object_list(box, src=0)
                plan32 = box[0]
            return plan32
        plan32 = build32(0 if n_ctx > 1 else -1)
        eng32 = make_engine(plan32)
        slots32 = make_slots(eng32, n_ctx)
        for k in range(2 * n_ctx):
            slots32[k % n_ctx].run(inputs[k % len(inputs)])
        tc.synchronize()
        dt32, legs32, _ = legs(slots32, n_side)
        one32 = make_slots(eng32, 1) if n_ctx > 1 else slots32
        dt32_1, legs32_1, _ = legs(one32, 3, warmup=min(args.warmup, 5)) if n_ctx > 1 else (dt32, legs32, None)
        tol = {"plan": plan32, "eng": eng32, "slots": slots32, "dt": dt32, "legs": legs32, "dt1": dt32_1, "legs1": legs32_1}

    # Host-fed variant (not `value`): what a caller pays when the boundary hands over HOST images, as the reference's demo does
    # (yolov8_det.cpp:146-160: cuda_batch_preprocess of cv::Mat frames, infer, D2H).  Raw uint8 HWC frames sit in pinned host
    # memory; a copy stream uploads batch k+1 while the slots run letterbox (preprocess.cu twin) -> enqueue -> NMS -> D2H of the
    # batches before it.  n_ctx + 1 upload buffers, event-fenced both ways.
    dt_host = host_legs = None
    if cfg["nms"]:
        from tensorrtx_amd import preproc
        n_up = args.upload_buffers or (2 * n_ctx + 2)   # a frame buffer is held until its batch's last kernel (the event the caller can see): with n_ctx + 1 of them the uploads waited for whole batches
        # the same synthetic scenes as the resident-input legs, as camera frames: uint8, HWC, BGR
        def as_frames(x):
            return pin(torch.from_numpy(np.ascontiguousarray((np.clip(x, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8).transpose(0, 2, 3, 1)[..., ::-1])))
        frames = [as_frames(rng_imgs[0])] * n_up if dry else [as_frames(rng_imgs[s % len(rng_imgs)]) for s in range(n_up)]
        raw = [torch.empty((batch, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(n_up)]
        net_in = [torch.empty((batch, 3, H, W), dtype=torch.float32, device=dev) for _ in range(n_ctx)]
        copy_stream = tc.Stream(priority=-1 if args.copy_priority else 0)   # a high-priority stream gets a hardware queue of its own pool: the uploads do not queue behind a context's kernels
        uploaded = [tc.Event() for _ in range(n_up)]
        consumed = [tc.Event() for _ in range(n_up)]

        def upload(s):
            with tc.stream(copy_stream):
                copy_stream.wait_event(consumed[s])      # the letterbox kernel of the previous user of this buffer is done
                raw[s].copy_(frames[s], non_blocking=True)
                uploaded[s].record(copy_stream)

        def host_step(k):
            # letterbox fused into the stem: the engine's first layer samples the uploaded uint8 frames itself (enqueue_frames); the fp32
            # network input of cuda_batch_preprocess is never written or read
            s, slot = k % n_up, slots[k % n_ctx]
            slot.stream.wait_event(uploaded[s])
            slot.run(None if not dry else net_in[k % n_ctx], with_d2h=True, frames=list(raw[s]))
            consumed[s].record(slot.stream)

        def host_timed(n_steps, uploads=True):
            for s in range(n_up):
                consumed[s].record(tc.current_stream())
            tc.synchronize()
            if dist:
                dist.barrier()
            tc.synchronize()
            t0 = time.perf_counter()
            if uploads:
                upload(0)
            for k in range(n_steps):
                if uploads and k + 1 < n_steps:
                    upload((k + 1) % n_up)               # prefetch the next batch while the earlier ones compute
                host_step(k)
            tc.synchronize()
            if dist:
                dist.barrier()
            tc.synchronize()
            return replicas.max_over_ranks(time.perf_counter() - t0, dist, dev)

        def h2d_alone(n_steps):
            # the uploads of the same frames with nothing else on the GPU: what the PCIe link itself sustains for this transfer size
            tc.synchronize()
            t0 = time.perf_counter()
            with tc.stream(copy_stream):
                for k in range(n_steps):
                    raw[k % n_up].copy_(frames[k % n_up], non_blocking=True)
            tc.synchronize()
            return time.perf_counter() - t0

        host_timed(args.steps)                           # warm the path: letterbox kernel, pinned copies, the link (round 5: the first leg after a 6-step warm-up ran 2x slow, legs 3-7 within 0.5 %)
        host_all = [host_timed(args.steps) for _ in range(7)]   # seven legs like `value` (round 4 took three and one of them was a 2x outlier)
        dt_host = sorted(host_all)[3]
        host_legs = [round(h / args.steps * 1e3, 4) for h in host_all]
        # the same steps with the frames already on the device (the `uploaded` events of the last leg stay signalled): what the fused letterbox + stem
        # costs against the resident fp32 input of `value` - the rest of the distance to `value` is the link
        frames_resident_ms = sorted(host_timed(args.steps, uploads=False) for _ in range(3))[1] / args.steps * 1e3
        h2d_alone(2)
        h2d_ms = sorted(h2d_alone(args.steps) for _ in range(3))[1] / args.steps * 1e3
        host_frame_bytes = int(frames[0].numel())

    # per-kernel timing with HIP events on the launch stream (IProfiler analogue): roofline of the dominant kernel
    prof_runs = 5

    def profile_convs(e, bindings):
        """mean over prof_runs of (sum of the MFMA conv launches, sum of all launches) in ms, launch count, last per-op rows, tactics"""
        c_ms = t_ms = k_ms = 0.0
        n = 0
        rows = []
        for _ in range(prof_runs):
            rows = e.profile(batch, bindings)
            conv = [r for r, o in zip(rows, low["ops"]) if (o["kind"] == "conv" and o.get("igemm")) or o["kind"] == "conv_group"]
            n = len(conv)
            c_ms += sum(r["ms"] for r in conv)
            t_ms += sum(r["ms"] for r in rows)
            # the dispatches' own begin -> end (hipExtLaunchKernelGGL start / stop events): only if every conv launch reported one
            k_ms += sum(r["kernel_ms"] for r in conv) if all(r.get("kernel_ms", -1) > 0 for r in conv) else float("nan")
        tac = e.tactics()
        moved = [t for t in tac if t["tactic"] != t["default"]]
        summary = {"convs_timed": len(tac), "moved_off_default": len(moved),
                   "default_sum_us": round(sum(t["default_us"] for t in tac if t["default_us"] > 0), 1),
                   "chosen_sum_us": round(sum(t["us"] for t in tac if t["us"] > 0), 1)}
        k_ms = k_ms / prof_runs
        if not (0.5 * c_ms / prof_runs < k_ms <= 1.02 * c_ms / prof_runs):   # NaN or implausible: not used
            k_ms = None
        return c_ms / prof_runs, t_ms / prof_runs, n, rows, tac, summary, k_ms

    conv_ms_events, tot_ms, n_conv, rows, tac, tactic_summary, conv_ms_kernel = profile_convs(eng, slots[0].bindings(inputs[0]))
    # The figure the roofline is priced with: the kernels' own durations where the runtime could record them (they agree with what
    # rocprofv3 --kernel-trace reports for the same launches); otherwise the interval between the stream events around each op.
    conv_ms = conv_ms_kernel if conv_ms_kernel else conv_ms_events
    tactic_summary["what"] = ("in-place timing of every MFMA convolution's launch configurations when the plan is built (runtime/tune.cpp; the choices and the measured times travel in the plan, deserialize only applies them); TRTX_TUNE=0 keeps every layer on its static default" +
                              ("; this engine was built with setMaxAuxStreams(0) = contexts in flight: it chooses among the work-efficient configurations only" if n_ctx > 1 else ""))
    single_prof = None
    if n_ctx > 1:
        conv1_ev, tot1_ms, n1, _, _, tac1, conv1_k = profile_convs(eng1, one[0].bindings(inputs[0]))
        single_prof = (conv1_k if conv1_k else conv1_ev, tot1_ms, n1, tac1)
        eng1.close()
    if args.dump_ops and rank == 0:
        json.dump(tac, open(args.dump_ops + ".tactics.json", "w"), indent=0)
        json.dump([dict(r, **{k: o.get(k) for k in ("cin", "cout", "k", "hw_in", "hw_out", "residual", "flops", "kernel")})
                   for r, o in zip(rows, low["ops"])], open(args.dump_ops, "w"), indent=0)
    # Dominant kernel family = the fused MFMA convolutions.  ALGORITHMIC bytes per launch = fp16 activations in + out
    # (+ residual) at this batch + the layer's packed weights once (DESIGN.md "Measurement"); duration = per-op HIP events.
    # YOLOv8n layers sit below the MFMA/HBM ridge (arithmetic intensity 16..290 FLOP/B against 312) -> bound "hbm";
    # ResNet-50 / RetinaFace / R-CNN are dominated by layers above it -> bound "mfma".  Both views are always printed.
    # (a grouped launch - 2..4 independent sibling convolutions in one dispatch, round 4 - is ONE launch priced on the bytes of all its members)
    launch_ops = [o for o in low["ops"] if (o["kind"] == "conv" and o.get("igemm")) or o["kind"] == "conv_group"]
    n_group = sum(1 for o in launch_ops if o["kind"] == "conv_group")
    convs_in_groups = sum(len(o["members"]) for o in launch_ops if o["kind"] == "conv_group")
    igemm_ops = [m for o in launch_ops for m in (o["members"] if o["kind"] == "conv_group" else [dict(o)])]
    for m in igemm_ops:
        m.setdefault("kind", "conv")
    alg_bytes = 0.0
    flop_per_step = 0.0
    for o in igemm_ops:
        nb = o.get("nfix") or batch              # images per launch = nb * nmul (nmul: RoIs per image in the R-CNN head)
        i8 = o.get("i8", [0, 0, 0])
        es_in, es_out, es_res = (1.0 if i8[0] else 2.0), (1.0 if i8[1] else 2.0), (1.0 if i8[2] else 2.0)
        up_c = o.get("up_c", 0)   # folded upsample: those input channels are read from a tensor a quarter the size
        act_b = (es_in * o["hw_in"][0] * o["hw_in"][1] * (o["cin"] - 0.75 * up_c) + o["hw_out"][0] * o["hw_out"][1] * o["cout"] * (es_out + (es_res if o["residual"] else 0.0)))
        alg_bytes += act_b * nb * o.get("nmul", 1) + es_in * o["cout"] * o["cin"] * o["k"][0] * o["k"][1]
        flop_per_step += o["flops"] * nb          # plan flops are per sample and already include nmul
    avg_launch_s = conv_ms * 1e-3 / max(n_conv, 1)
    achieved_gbps = alg_bytes / max(n_conv, 1) / avg_launch_s / 1e9
    achieved_tflops = flop_per_step / (conv_ms * 1e-3) / 1e12
    hbm_view = {"bytes_per_launch": alg_bytes / max(n_conv, 1), "achieved_GBps": achieved_gbps, "peak_GBps": HBM_PEAK_GBPS,
                "frac": achieved_gbps / HBM_PEAK_GBPS}
    mfma_view = {"flop_per_launch": flop_per_step / max(n_conv, 1), "achieved_TFLOPs": achieved_tflops, "peak_TFLOPs": MFMA_PEAK_TFLOPS,
                 "frac": achieved_tflops / MFMA_PEAK_TFLOPS}
    intensity = flop_per_step / max(alg_bytes, 1.0)
    bound = "hbm" if intensity < MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBPS else "mfma"
    traffic, traffic_src = _traffic_from_profile(args.config)
    prof_us, prof_src = _kernel_duration_from_profile(args.config) if args.precision == "fp16" else (None, None)
    roofline = {"bound": bound, "kernel": "fused MFMA convolution kernels (conv_igemm_f16 / conv_igemm_group_f16 / conv_ws_f16 / conv_patch_f16 / conv_res3_f16 / conv_res1_f16 / conv_gemm256_f16 / conv_igemm_wsk_f16, all instantiations)",
                "launches_per_step": n_conv, "grouped_launches": n_group, "convolutions_inside_groups": convs_in_groups,
                "bytes_priced": "algorithmic: fp16 activations in + out (+ residual) of every launch + its weights once", "avg_launch_us": avg_launch_s * 1e6,
                "timing": ("dispatch begin -> end of every conv launch (HIP events attached to the launch, hipExtLaunchKernelGGL), mean of 5 serialized profile passes"
                           if conv_ms_kernel else "interval between the HIP stream events around every conv op, mean of 5 serialized profile passes"),
                "avg_launch_us_between_stream_events": conv_ms_events * 1e3 / max(n_conv, 1),
                "achieved": achieved_gbps if bound == "hbm" else achieved_tflops, "peak": HBM_PEAK_GBPS if bound == "hbm" else MFMA_PEAK_TFLOPS,
                "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                "frac": (achieved_gbps / HBM_PEAK_GBPS) if bound == "hbm" else (achieved_tflops / MFMA_PEAK_TFLOPS),
                "traffic": traffic, "traffic_source": traffic_src,
                "rocprofv3_avg_launch_us": prof_us, "rocprofv3_frac_hbm": (alg_bytes / max(n_conv, 1) / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if prof_us else None,
                "rocprofv3_source": prof_src,
                "rocprofv3_like_for_like": "single_context.roofline.avg_launch_us: the committed profile is of the ONE-context engine on one lane (its own tactic set), the figure above of the engine that produced `value`",
                "arithmetic_intensity_flop_per_byte": intensity,
                "conv_ms_per_step": conv_ms, "all_kernels_ms_per_step": tot_ms, "hbm_view": hbm_view, "mfma_view": mfma_view,
                "tactics": tactic_summary,
                "frac_describes": ("KERNEL QUALITY, not the timed configuration: every conv launch timed alone in a serialized one-stream profile pass of the engine that "
                                   "produced `value`; the timed region keeps several batches in flight, so its launches overlap - what the whole step sustains is "
                                   "whole_step_hbm_view.frac"),
                "whole_step_hbm_view": {"algorithmic_bytes_per_step": alg_bytes, "GBps_at_measured_step": alg_bytes / (dt / args.steps) / 1e9,
                                        "frac": alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS}}
    # ---- the plugin path on the roofline ("decode / NMS / top-k / RoIAlign / reformat -> HBM bandwidth").  Every launch of
    # the step that is NOT a convolution: algorithmic bytes (what the lowering pass states for the op: its inputs + outputs once, at this batch), the interval
    # between the stream events around the op in the same serialized profile passes as above, GB/s, fraction of the HBM peak.  Ops of a few hundred KB
    # are latency, not bandwidth: their rows say so.  The GPU NMS (outside the engine: trtx_yolo_nms on the decode buffer) is timed the same way.
    DT_SIZE = {0: 4, 1: 2, 2: 1}
    tens = {t["id"]: t for t in low.get("tensors", [])}

    def op_bytes(o):
        if o.get("bytes", 0) > 0:
            return o["bytes"] * batch
        tot = 0.0
        for tid in list(o.get("in", [])) + list(o.get("out", [])):
            t = tens.get(tid)
            if t:
                tot += float(np.prod(t["dims"])) * DT_SIZE.get(t.get("dtype", 0), 4) * batch
        return tot

    plugins = []
    if rows and len(rows) == len(low["ops"]):
        acc = {}
        for r, o in zip(rows, low["ops"]):
            if (o["kind"] == "conv" and o.get("igemm")) or o["kind"] == "conv_group":
                continue
            key = o["kind"] if o["kind"] != "plugin" else "plugin:" + o.get("name", "")[:40]
            a = acc.setdefault(key, {"op": key, "launches_per_step": 0, "bytes": 0.0, "us": 0.0})
            a["launches_per_step"] += 1
            a["bytes"] += op_bytes(o)
            a["us"] += r["ms"] * 1e3
        for a in acc.values():
            gbps = a["bytes"] / (a["us"] * 1e-6) / 1e9 if a["us"] > 0 else 0.0
            plugins.append({"op": a["op"], "launches_per_step": a["launches_per_step"], "algorithmic_bytes_per_step": a["bytes"], "us_per_step": round(a["us"], 2),
                            "GBps": round(gbps, 1), "frac_hbm": round(gbps / HBM_PEAK_GBPS, 4),
                            "bound": "hbm" if a["bytes"] / max(a["launches_per_step"], 1) > 8e6 else "latency (under 8 MB per launch: the ~6 us floor of a launch is more than its bytes)"})
    if cfg["nms"] and not dry:
        sl0 = slots[0]
        sl0.run(inputs[0])
        tc.synchronize()
        cnt = sl0.keep_cnt.cpu().numpy()
        n_cand = sl0.out[:, 0].cpu().numpy().clip(0, 1000)
        ev = [tc.Event(enable_timing=True) for _ in range(2)]
        nms_us = []
        for _ in range(7):
            with tc.stream(sl0.stream):
                ev[0].record()
            capi.check(L.trtx_yolo_nms(capi._p(sl0.out), batch, 1000, ctypes.c_float(0.5), ctypes.c_float(0.45), capi._p(sl0.keep_idx), capi._p(sl0.keep_cnt),
                                       capi._p(sl0.keep_det), capi._p(sl0.ws), ctypes.c_size_t(sl0.ws_bytes), sl0.stream_p), "trtx_yolo_nms")
            with tc.stream(sl0.stream):
                ev[1].record()
            tc.synchronize()
            nms_us.append(ev[0].elapsed_time(ev[1]) * 1e3)
        nms_b = float(n_cand.sum()) * 24.0 + float(cnt.sum()) * 28.0 + batch * 8.0   # 6 floats read per candidate, index + 6 floats written per kept box, the counts
        us = sorted(nms_us)[len(nms_us) // 2]
        plugins.append({"op": "trtx_yolo_nms (yolo_nms_sort + yolo_nms_mask + yolo_nms_scan, after the engine)", "launches_per_step": 3, "algorithmic_bytes_per_step": nms_b,
                        "us_per_step": round(us, 2), "GBps": round(nms_b / (us * 1e-6) / 1e9, 2), "frac_hbm": round(nms_b / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 6),
                        "candidates_per_image": float(n_cand.mean()), "kept_per_image": float(cnt.mean()),
                        "bound": "latency: three dependent one-workgroup-per-image phases (1024-wide bitonic sort 15 us of it, profiles/r06_nms_fused_ab.txt)"})
    roofline["plugins"] = plugins
    roofline["plugins_note"] = ("every non-convolution launch of the step, same serialized profile passes (stream events around the op); per-kernel rocprofv3 durations of the same "
                                "command: profiles/r06_kernel_stats_*_1ctx_lanes1.txt")
    # the three fractions side by side: every conv launch alone (serialized), the one-context engine's, the whole timed step's
    roofline["frac_serialized_kernels"] = roofline["frac"]
    roofline["frac_whole_step_hbm"] = roofline["whole_step_hbm_view"]["frac"]
    roofline["frac_single_context_hbm"] = (alg_bytes / (single_prof[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if single_prof else None
    res = {
        "metric": f"images/sec @ batch={cfg['batch']} {W}x{H} {args.precision} ({args.config}" + (" conv backbone + YoloLayer decode + NMS)" if cfg["nms"] else ", IExecutionContext::enqueue)"),
        "value": (global_batch if mode == "strong" else world * batch) * args.steps / dt, "unit": "images/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": mode,
        "legs_ms": value_legs,
        "timing": (f"{len(value_legs)} legs back to back, each = {args.warmup} untimed warm-up steps + exactly {args.steps} timed steps between barrier + "
                   f"synchronize on both sides (max over ranks); `value` / `ms_per_step` are the MEDIAN leg, `legs_ms` lists all of them "
                   f"(ms per step); {settle_steps} untimed settle steps ran before the first leg"),
        "vs_baseline": None, "dtype": "f16" if args.precision == "fp16" else "i8 (+f16 fallback layers)", "data": "synthetic",
        "config": {"workload": f"{args.config} {args.precision} {W}x{H}, {cfg['tag']}: per-GPU batch {batch}, " +
                               ("enqueue + GPU NMS" if cfg["nms"] else "enqueue (all plugins inside the engine)") +
                               (f", {n_ctx} execution contexts in flight (one stream each, setMaxAuxStreams(0)); every step is a full batch-{batch} pass"
                                if n_ctx > 1 else ", one execution context (3 auxiliary streams), steps strictly in sequence") +
                               f", {len(inputs)} rotating input batches resident in HBM",
                   "contexts": n_ctx,
                   "global_batch": global_batch, "parallelism": f"replica-per-GPU x{world} (image-sharded, no data-path collective; RCCL only brackets the timed region)",
                   "weights": "seeded synthetic .wts (no trained weights offline)", "cpu_affinity_rank0": affinity_note, "cpu_affinity_per_rank": affinity_all},
        "per_rank_ms_per_step": value_trace.get("per_rank_ms_per_step_median_leg"),
        "roofline": roofline,
    }
    if dt_single is not None:
        res["single_context"] = {"value": (global_batch if mode == "strong" else world * batch) * args.steps / dt_single, "unit": "images/sec",
                                 "ms_per_step": dt_single / args.steps * 1e3, "legs_ms": single_legs,
                                 "what": "the same K steps through ONE execution context, strictly one batch after the other (the shape of the reference's loop, yolov8_det.cpp:97-104): the per-batch latency figure; this engine uses 3 auxiliary streams and the full tactic set"}
        c1_ms, t1_ms, n1, tac1 = single_prof
        res["single_context"]["roofline"] = {"avg_launch_us": c1_ms * 1e3 / max(n1, 1), "conv_ms_per_step": c1_ms, "all_kernels_ms_per_step": t1_ms,
                                             "hbm_frac": alg_bytes / (c1_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "mfma_frac": flop_per_step / (c1_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                             "tactics": tac1}
    if tol is not None:
        imgs_step = (global_batch if mode == "strong" else world * batch)
        low32 = engine.describe_plan(tol["plan"], lowered=True)
        c32 = k32 = 0.0
        rows32 = []
        if not dry:
            for _ in range(3):
                rows32 = tol["eng"].profile(batch, tol["slots"][0].bindings(inputs[0]))
                conv32 = [r for r, o in zip(rows32, low32["ops"]) if o["kind"] == "conv" and o.get("igemm")]
                c32 += sum(r["ms"] for r in conv32) / 3
                k32 += (sum(r["kernel_ms"] for r in conv32) if all(r.get("kernel_ms", -1) > 0 for r in conv32) else float("nan")) / 3
        conv32_ms = k32 if (k32 == k32 and 0.5 * c32 < k32 <= 1.02 * c32) else c32
        flop32 = sum(o["flops"] for o in low32["ops"] if o["kind"] == "conv" and o.get("igemm")) * batch
        n32 = sum(1 for o in low32["ops"] if o["kind"] == "conv" and o.get("igemm"))
        res["tolerance_engine"] = {
            "what": ("the same network built WITHOUT kFP16 (the reference's USE_FP32 build, yolov8/include/config.h:1-3): fp32 storage, every convolution on "
                     "v_mfma_f32_16x16x4_f32 (exact fp32 products and sums, two-level K sum), fp32 stem kernel, fused DFL + decode on fp32 head tensors; "
                     "timed like `value` (same contexts in flight, inputs resident, decode + NMS included)"),
            "value": imgs_step * args.steps / tol["dt"], "unit": "images/sec", "ms_per_step": tol["dt"] / args.steps * 1e3, "legs_ms": tol["legs"],
            "dtype": "f32", "contexts": n_ctx,
            "single_context": {"value": imgs_step * args.steps / tol["dt1"], "ms_per_step": tol["dt1"] / args.steps * 1e3, "legs_ms": tol["legs1"]},
            "roofline": {"bound": "mfma", "kernel": "conv_igemm_f16_kernel<..., F32> + conv_res3 / conv_res1 <..., F32> (kernels/conv_igemm_f32.hip, conv_res.hip: every instantiation the tuner picked)", "launches_per_step": n32,
                         "flop_per_step": flop32, "conv_ms_per_step": conv32_ms, "achieved": flop32 / (conv32_ms * 1e-3) / 1e12 if conv32_ms else None,
                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop32 / (conv32_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if conv32_ms else None,
                         "frac_describes": "every fp32 conv launch alone (serialized profile pass), against the fp32-operand MFMA peak (1/16 of the fp16 one)"},
        }
    if cfg["nms"]:
        res["d2h_inclusive"] = {"value": (global_batch if mode == "strong" else world * batch) * args.steps / dt_d2h, "unit": "images/sec",
                                "ms_per_step": dt_d2h / args.steps * 1e3, "legs_ms": d2h_legs,
                                "what": "same steps + async copy of the kept counts and the compacted detection buffer [B,1000,6] to pinned host memory each step (the reference's timer includes D2H, yolov8_det.cpp:97-104)"}
        res["host_fed"] = {"value": (global_batch if mode == "strong" else world * batch) * args.steps / dt_host, "unit": "images/sec",
                           "ms_per_step": dt_host / args.steps * 1e3, "legs_ms": host_legs,
                           "pcie": {"h2d_bytes_per_step": host_frame_bytes, "h2d_alone_ms_per_step": h2d_ms, "h2d_alone_GBps": host_frame_bytes / (h2d_ms * 1e-3) / 1e9 if h2d_ms else None,
                                    "h2d_share_of_step": h2d_ms / (dt_host / args.steps * 1e3) if dt_host else None, "frames_resident_ms_per_step": frames_resident_ms,
                                    "what": "the uploads of the same pinned frames alone (no kernels): the link's own time per step - the host-fed rate is bounded by max(upload, compute)"},
                           "what": "PCIe-inclusive: uint8 HWC frames in pinned host memory -> H2D on a copy stream (double-buffered, overlapped) -> enqueue_frames (the letterbox is fused into the stem convolution: no fp32 input tensor) -> NMS -> D2H of the detections; never `value`"}
        res["detections"] = detections
    # internal consistency: a leg that does strictly MORE work per step (the D2H copies) or keeps fewer batches in flight (one
    # context) cannot be faster than `value`, and the legs of `value` should agree with each other; if not, say so in the line
    med_ms = dt / args.steps * 1e3
    why = []
    if dt_d2h is not None and dt_d2h < 0.97 * dt:
        why.append("d2h_inclusive is faster than value")
    if dt_single is not None and dt_single < 0.97 * dt:
        why.append("single_context is faster than value")
    if max(value_legs) > 1.25 * med_ms:
        why.append(f"slowest value leg {max(value_legs):.3f} ms/step vs median {med_ms:.3f}")
    res["suspect"] = bool(why)
    if why:
        res["suspect_why"] = why
    if max(value_legs) > 1.25 * med_ms:
        res["slowest_leg_trace"] = value_trace   # per step: when the host's enqueue returned / when the device finished it (ms)
    if dry:
        res["dry_run"] = True
        res["data"] = "NONE: --dry-run rehearses the control flow on CPU; every number in this line is meaningless"
    if rank == 0:
        if args.config == "yolov8n" and not args.no_cpu_baseline and not dry:
            # GPU outputs for the oracle's sample images (a separate small engine run, outside every timed region)
            nb = 4
            if args.precision == "int8":   # the scales the benchmarked engine was calibrated to
                from tensorrtx_amd import calibrator
                with calibrator.Calibrator(cache=calib_cache["text"]).installed():
                    plan_h = engine.build_plan("yolov8n", path, batch=nb, h=H, w=W, fp16=1, int8=1, mark_heads=1)
            else:
                plan_h = engine.build_plan("yolov8n", path, batch=nb, h=H, w=W, fp16=1, mark_heads=1)
            eh = engine.Engine(plan_h)
            imgs = synth.images(nb, H, W, seed=11)
            bufs = [torch.from_numpy(imgs).to(dev)] + [torch.empty(nb * int(np.prod(eh.dims[i])), dtype=torch.float32, device=dev)
                                                       for i in range(1, eh.nb_bindings)]
            eh.enqueue(nb, bufs)
            tc.synchronize()
            heads = [bufs[eh.names.index(f"head{i}")].cpu().numpy().reshape(nb, -1) for i in range(3)]
            eh.close()
            # the decoded boxes come from the TIMED engine itself (batch 32, fused head, folded upsample, grouped launches): its own batch
            # with the oracle's sample images in front (VERDICT r3 Weak 4: the line used to compare a separate batch-4 plan only)
            mixed = np.concatenate([imgs, rng_imgs[0][nb:]], 0) if batch > nb else imgs[:batch]
            s0 = slots[0]
            s0.ctx.enqueue(batch, s0.bindings(torch.from_numpy(np.ascontiguousarray(mixed)).to(dev)), stream=s0.stream.cuda_stream)
            tc.synchronize()
            dec = s0.outs[eng.names.index("output")].cpu().numpy().reshape(batch, -1)[:nb]
            extra = None
            if tol is not None:
                e32h = engine.Engine(engine.build_plan("yolov8n", path, batch=nb, h=H, w=W, fp16=0, mark_heads=1))
                b32 = [torch.from_numpy(imgs).to(dev)] + [torch.empty(nb * int(np.prod(e32h.dims[i])), dtype=torch.float32, device=dev) for i in range(1, e32h.nb_bindings)]
                e32h.enqueue(nb, b32)
                tc.synchronize()
                heads32 = [b32[e32h.names.index(f"head{i}")].cpu().numpy().reshape(nb, -1) for i in range(3)]
                e32h.close()
                t0s = tol["slots"][0]
                t0s.ctx.enqueue(batch, t0s.bindings(torch.from_numpy(np.ascontiguousarray(mixed)).to(dev)), stream=t0s.stream.cuda_stream)
                tc.synchronize()
                dec32 = t0s.outs[tol["eng"].names.index("output")].cpu().numpy().reshape(batch, -1)[:nb]
                extra = {"tolerance_engine": (heads32, dec32, "fp32 build (fp32 storage, fp32 MFMA)")}
            res["cpu_baseline"], res["parity"] = cpu_baseline_and_parity(path, heads, dec, imgs, precision=args.precision, extra=extra)
            if tol is not None:
                res["tolerance_engine"]["parity"] = res["parity"].pop("extra")["tolerance_engine"]
                res["tolerance_engine"]["parity"]["boxes_from"] = "the TIMED fp32 engine (batch %d, production plan)" % batch
            res["parity"]["boxes_from"] = "the TIMED engine (batch %d, production plan: fused head, folded upsample, grouped launches)" % batch
            res["parity"]["head_logits_from"] = "a batch-4 plan of the same network with the three head tensors marked as outputs (the timed plan fuses them into the decode kernel)"
        elif not args.no_cpu_baseline and not dry:
            nb = {"resnet50": 8, "retinaface_r50": 1, "rcnn_r50c4": 1}[args.config]
            s0 = slots[0]
            par_in, par_np = inputs[0], rng_imgs[0]
            if args.config == "retinaface_r50":
                # the timed batches are raw 0..255 pixels (what the reference feeds after its mean subtraction, retina_r50.cpp:259-262, is of that
                # size); with the seeded random weights that range drives the box regressions through exp() to inf - fine for timing, useless
                # for comparing boxes.  The parity leg therefore runs the SAME timed engine on a normalised batch (the tests' (255 x - 110) / 64).
                par_np = ((synth.images(batch, H, W, seed=2) * 255.0 - 110.0) / 64.0).astype(np.float32)
                par_in = torch.from_numpy(par_np).to(dev)
            s0.ctx.enqueue(batch, s0.bindings(par_in), stream=s0.stream.cuda_stream)
            tc.synchronize()
            gpu_out = {eng.names[i]: t.cpu().numpy() for i, t in s0.outs.items()}
            sample = par_np[:nb]
            if nhwc_input:
                sample = np.ascontiguousarray(sample.transpose(0, 2, 3, 1))
            res["cpu_baseline"], res["parity"] = cpu_baseline_and_parity_other(args.config, path, H, W, np.ascontiguousarray(sample, dtype=np.float32), gpu_out)
        print(json.dumps(res), flush=True)
    if tol is not None:
        tol["eng"].close()
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
