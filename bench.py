#!/usr/bin/env python3
"""Headline benchmark: images/sec of the YOLOv8n detection hot path (fp16 conv backbone -> YoloLayer decode
-> NMS) at batch 32, 640x640, on N MI355X GPUs of one node (BASELINE.json metric; config C3).

A "step" = one pass of the hot path over one batch of 32 synthetic images already resident in HBM:
IExecutionContext::enqueue (63 fused MFMA convolutions + pools/resizes + DFL tail + YoloLayer plugin)
followed by the GPU NMS.  Independent image streams are sharded over the ranks (one process per GPU, one
engine replica each, no data-path collective) -> weak scaling.

Prints ONE JSON line (rank 0) with the driver contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 32
SIZE = 640
GFLOP_PER_IMAGE = 8.743          # conv FLOP of YOLOv8n @640 (2*MAC), re-derived by the lowering pass
MFMA_PEAK_TFLOPS = 2500.0        # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming copy reaches)
# HBM traffic of the conv kernels per launch, measured in a separate `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum`
# pass over this same command (profiles/r01_pmc_conv_traffic.txt): (2 x RDREQ + WRREQ) x 64 B summed over the conv
# launches of a step / launches (reads doubled per the gfx950 note in MI355X_MICROARCH.md "HBM").  None = not measured.
PMC_TRAFFIC_BYTES_PER_LAUNCH = 39352748  # profiles/r01_pmc_conv_traffic.txt (algorithmic: 34.46 MB -> 1.14x)


def cpu_baseline(path, seconds_budget=20.0):
    """The oracle (PyTorch-CPU fp32 restatement of the reference graph + C decode/NMS) timed on the host cores
    on a bounded sample of the same workload.  Reported next to the GPU number; never the thing measured."""
    import numpy as np
    import torch

    from oracle import models_torch as mt
    from oracle import wts as owts
    from oracle import yolo_post as yp
    from tensorrtx_amd import synth

    # small-channel convolutions stop scaling (and oversubscribe badly) past a few dozen threads
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    params = owts.load_wts(path)
    nb = 4
    x = torch.from_numpy(synth.images(nb, SIZE, SIZE, seed=11))

    def once():
        with torch.inference_mode():
            heads, strides = mt.yolov8_det(mt.Params(params), x)
        dec = yp.decode_c([h.numpy() for h in heads], 80, SIZE, SIZE, strides)
        yp.batch_nms_c(dec)

    once()  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        once()
        n += nb
        if time.perf_counter() - t0 > seconds_budget or n >= 64:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} images ({nb}/iter) 640x640 fp32, PyTorch-CPU restatement of the reference graph + C decode/NMS"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-ops", default="", help="write the per-op hipEvent timing table (JSON) to this path")
    args = ap.parse_args()

    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from tensorrtx_amd import capi, engine, replicas, synth
    from tensorrtx_amd import wts as wts_writer

    # seeded synthetic YOLOv8n weights (no trained weights offline), written once per box in the reference's .wts format
    path = os.path.join(os.environ.get("TRTX_TEST_CACHE", "/tmp/trtx_test_cache"), "bench_yolov8n_seed0.wts")
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        wts_writer.write_wts(tmp, synth.yolov8n_state(seed=0), dialect="double")
        os.replace(tmp, path)
    plan = engine.build_plan("yolov8n", path, batch=BATCH, h=SIZE, w=SIZE, fp16=1)
    low = engine.describe_plan(plan, lowered=True)
    eng = engine.Engine(plan)
    x = torch.from_numpy(synth.images(BATCH, SIZE, SIZE, seed=100 + rank)).to(dev)  # resident in HBM
    out = torch.empty((BATCH, 1 + 1000 * 90), dtype=torch.float32, device=dev)
    bindings = [x, out]
    L = capi.lib()
    keep_idx = torch.empty((BATCH, 1000), dtype=torch.int32, device=dev)
    keep_cnt = torch.empty((BATCH,), dtype=torch.int32, device=dev)
    keep_det = torch.empty((BATCH, 1000, 6), dtype=torch.float32, device=dev)
    import ctypes
    stream = capi._stream()
    L.trtx_yolo_nms_workspace.restype = ctypes.c_size_t
    nms_ws_bytes = L.trtx_yolo_nms_workspace(BATCH)
    nms_ws = torch.empty((nms_ws_bytes,), dtype=torch.uint8, device=dev)

    def step():
        eng.enqueue(BATCH, bindings)
        capi.check(L.trtx_yolo_nms(capi._p(out), BATCH, 1000, ctypes.c_float(0.5), ctypes.c_float(0.45), capi._p(keep_idx),
                                   capi._p(keep_cnt), capi._p(keep_det), capi._p(nms_ws), ctypes.c_size_t(nms_ws_bytes), stream),
                   "trtx_yolo_nms")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = replicas.max_over_ranks(dt, dist, dev)

    # per-kernel timing with HIP events on the launch stream (IProfiler analogue): roofline of the dominant kernel
    prof_runs = 5
    conv_ms = 0.0
    tot_ms = 0.0
    n_conv = 0
    for _ in range(prof_runs):
        rows = eng.profile(BATCH, bindings)
        conv = [r for r, o in zip(rows, low["ops"]) if o["kind"] == "conv" and o.get("igemm")]
        n_conv = len(conv)
        conv_ms += sum(r["ms"] for r in conv)
        tot_ms += sum(r["ms"] for r in rows)
    if args.dump_ops and rank == 0:
        json.dump(rows, open(args.dump_ops, "w"), indent=0)
    conv_ms /= prof_runs
    tot_ms /= prof_runs
    # Dominant kernel = the fused implicit-GEMM convolution (62 launches per step).  Every YOLOv8n layer sits below the
    # MFMA/HBM ridge (arithmetic intensity 16..290 FLOP/B against 312), so the bound that applies is HBM: achieved =
    # ALGORITHMIC bytes per launch (fp16 activations in + out (+ residual) at batch 32, + the layer's packed weights once;
    # DESIGN.md "Measurement") / average launch duration from the per-op HIP events above.  The MFMA view is reported too.
    igemm_ops = [o for o in low["ops"] if o["kind"] == "conv" and o.get("igemm")]
    alg_bytes = 0.0
    for o in igemm_ops:
        act = o["hw_in"][0] * o["hw_in"][1] * o["cin"] + o["hw_out"][0] * o["hw_out"][1] * o["cout"] * (2 if o["residual"] else 1)
        alg_bytes += 2.0 * act * BATCH + 2.0 * o["cout"] * o["cin"] * o["k"][0] * o["k"][1]
    flop_per_step = sum(o["flops"] for o in igemm_ops) * BATCH
    avg_launch_s = conv_ms * 1e-3 / max(n_conv, 1)
    achieved_gbps = alg_bytes / max(n_conv, 1) / avg_launch_s / 1e9
    achieved_tflops = flop_per_step / (conv_ms * 1e-3) / 1e12
    roofline = {"bound": "hbm", "kernel": "conv_igemm_f16_kernel / conv_igemm_wsk_f16_kernel (fused implicit-GEMM conv, all instantiations)",
                "launches_per_step": n_conv, "avg_launch_us": avg_launch_s * 1e6,
                "bytes_per_launch": alg_bytes / max(n_conv, 1), "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved_gbps / HBM_PEAK_GBPS,
                # HBM bytes per launch from the L2 fabric counters (separate rocprofv3 --pmc pass, gfx950 read correction
                # applied): see PMC_TRAFFIC_BYTES_PER_LAUNCH
                "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH,
                "conv_ms_per_step": conv_ms, "all_kernels_ms_per_step": tot_ms,
                "mfma_view": {"flop_per_launch": flop_per_step / max(n_conv, 1), "achieved_TFLOPs": achieved_tflops,
                              "peak_TFLOPs": MFMA_PEAK_TFLOPS, "frac": achieved_tflops / MFMA_PEAK_TFLOPS},
                "whole_step_hbm_view": {"algorithmic_bytes_per_step": alg_bytes,
                                        "GBps_at_measured_step": alg_bytes / (dt / args.steps) / 1e9}}
    res = {
        "metric": "images/sec @ batch=32 640x640 fp16 (YOLOv8n conv backbone + YoloLayer decode + NMS)",
        "value": world * BATCH * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "yolov8n fp16 640x640 batch=32 per GPU (BASELINE configs[2]): enqueue + GPU NMS, inputs resident in HBM",
                   "global_batch": world * BATCH, "parallelism": f"replica-per-GPU x{world} (image-sharded, no collective)",
                   "weights": "seeded synthetic .wts (no trained weights offline)"},
        "roofline": roofline,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(path)
        print(json.dumps(res), flush=True)
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
