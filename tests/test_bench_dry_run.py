"""bench.py's control flow on CPU: the exact launch line the driver uses for N > 1 (python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N --steps K --warmup W), with --dry-run putting stand-ins in place of every GPU object and gloo in place of RCCL.
What runs for real: RANK / WORLD_SIZE handling, rank 0 building the plan and broadcasting it, the per-rank input seeds, settle / warm-up
/ K-step legs with their barriers and the max-over-ranks reduce, the host-fed ring, one JSON line from rank 0 only."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n, extra=()):
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "1", "--steps", "6", "--warmup", "2", "--repeats", "3", *extra]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port",
               str(_free_port()), os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", str(n), "--steps", "6", "--warmup", "2", "--repeats", "3", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line (rank 0 only), got {len(lines)}"
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2])
def test_bench_control_flow_weak_scaling(n):
    d = _run(n)
    assert d["dry_run"] is True and d["n_gpus"] == n and d["steps"] == 6 and d["warmup"] == 2
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "images/sec"
    assert len(d["legs_ms"]) == 3 and all(x > 0 for x in d["legs_ms"])
    assert abs(d["value"] - 32 * n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]      # whole-job aggregate over all ranks
    assert d["config"]["workload"] and "roofline" in d and d["roofline"]["launches_per_step"] > 40
    # every rank's own step time and pinning report travel in the line: a straggler is visible, not folded into the max (VERDICT r5 Weak 10)
    assert len(d["per_rank_ms_per_step"]) == n and all(0 < x <= d["ms_per_step"] * (1 + 1e-6) + 1e-3 for x in d["per_rank_ms_per_step"])
    assert len(d["config"]["cpu_affinity_per_rank"]) == n
    assert "plugins" in d["roofline"] and "frac_serialized_kernels" in d["roofline"] and "frac_whole_step_hbm" in d["roofline"]
    assert ("cpu_baseline" in d) == False   # noqa: E712  (the oracle needs real GPU outputs to be compared with: not in a dry run)


def test_bench_control_flow_strong_scaling_two_ranks():
    d = _run(2, ("--mode", "strong"))
    assert d["scaling"] == "strong" and d["n_gpus"] == 2
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # the global batch is fixed: 16 + 16


def test_bench_control_flow_c4_strong_split_builds_small_batch_engines():
    """C4 (RetinaFace-R50 1280x1280, global batch 8) is strong-scaled by default: with 8 GPUs every rank gets ONE image, so the builders must
    accept a max batch below the config's.  Two ranks here (4 images each) through the same code; the batch-1 build itself is checked on
    this process (the shape rank k of 8 would build)."""
    d = _run(2, ("--config", "retinaface_r50"))
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["config"]["global_batch"] == 8
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tensorrtx_amd import engine, replicas
    from util import synth_wts
    assert [len(list(replicas.partition(8, 8, r))) for r in range(8)] == [1] * 8
    path, _ = synth_wts("retinaface_r50")
    plan = engine.build_plan("retinaface_r50", path, batch=1, fp16=1, h=1280, w=1280, aux_streams=0)
    assert engine.describe_plan(plan, lowered=True)["max_batch"] == 1


def test_bench_control_flow_c4_eight_ranks_one_image_each():
    """the driver's 8-GPU line for C4 - `--gpus 8 --config retinaface_r50`, strong scaling by default: rank k gets image k of the global
    batch of 8 - rehearsed with eight gloo ranks: plan broadcast of a batch-1 engine, per-rank partition, barriers, max over ranks, one JSON line."""
    d = _run(8, ("--config", "retinaface_r50"))
    assert d["scaling"] == "strong" and d["n_gpus"] == 8 and d["config"]["global_batch"] == 8
    assert len(d["per_rank_ms_per_step"]) == 8 and len(d["config"]["cpu_affinity_per_rank"]) == 8
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_numa_pinning_helpers_without_a_gpu():
    """replicas.pin_to_gpu_numa_node: cpulist parsing, and no GPU / no sysfs entry leaves the affinity alone and says why (never raises)."""
    from tensorrtx_amd import replicas
    assert replicas.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and replicas.parse_cpulist("5") == [5]
    before = os.sched_getaffinity(0)
    note = replicas.pin_to_gpu_numa_node(0, 0, 1)
    assert note.startswith("affinity unchanged") and os.sched_getaffinity(0) == before


def test_bench_prints_the_tolerance_engine_leg():
    """the default line (yolov8n, fp16) carries `tolerance_engine` - the fp32 build timed like `value` - with its own legs, dtype and
    roofline against the fp32 MFMA peak; --no-tolerance-engine drops it."""
    d = _run(1)
    t = d["tolerance_engine"]
    assert t["dtype"] == "f32" and t["value"] > 0 and len(t["legs_ms"]) >= 3 and t["roofline"]["peak"] == 157.3 and t["roofline"]["launches_per_step"] >= 60
    assert "tolerance_engine" not in _run(1, ("--no-tolerance-engine",))


def test_bench_in_process_replicas_control_flow():
    """--replicas in-process: one process, N devices through DeviceReplicas (the reference's tutorials/multi_GPU_processing.md recipe); dry run
    with three stand-in devices: every device gets a batch per step, value = all devices' images over the host clock."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--replicas", "in-process", "--gpus", "3", "--steps", "6", "--warmup", "2",
                        "--repeats", "3"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dry_run"] is True and d["n_gpus"] == 3 and d["config"]["global_batch"] == 96 and len(d["legs_ms"]) == 3
    assert abs(d["value"] - 96 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
