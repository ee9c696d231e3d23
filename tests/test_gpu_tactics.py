"""Tactic selection by measurement (runtime/tune.cpp; what TensorRT's builder does with its tactics behind
IBuilder::buildSerializedNetwork, yolov8/src/model.cpp:327).  The timing runs when the plan is BUILT on a machine with a GPU and the
choices travel IN the plan (format 4): deserializeCudaEngine only applies them, so a plan file runs the same kernels - and returns the
same bits - in every process.  Every MFMA convolution gets a record; engines built for contexts in flight (setMaxAuxStreams(0)) choose
among the work-efficient configurations only; TRTX_TUNE=0 keeps the static defaults at build and at deserialize."""
import numpy as np
import pytest
import torch

from tensorrtx_amd import engine, synth
from util import synth_wts

pytestmark = pytest.mark.gpu


def _run(e, x, gpu):
    bufs = [x.to(gpu) if e.is_input[i] else torch.zeros(x.shape[0] * int(np.prod(e.dims[i])), dtype=torch.float32, device=gpu)
            for i in range(e.nb_bindings)]
    e.enqueue(x.shape[0], bufs)
    torch.cuda.synchronize()
    return {e.names[i]: bufs[i].cpu() for i in range(e.nb_bindings) if not e.is_input[i]}


def test_tactics_are_recorded_stable_and_change_nothing_but_speed(gpu, monkeypatch):
    monkeypatch.delenv("TRTX_TUNE", raising=False)
    path, _ = synth_wts("yolov8n")
    B, S = 6, 288   # a shape no other test builds: the process-wide choice cache is empty for it
    plan = engine.build_plan("yolov8n", path, batch=B, h=S, w=S, fp16=1, mark_heads=1)
    desc = engine.describe_plan(plan)
    assert desc["tactics_timed"] and desc["tactics"] >= 20       # built next to a GPU: the plan carries the choices
    assert engine.build_plan("yolov8n", path, batch=B, h=S, w=S, fp16=1, mark_heads=1) == plan   # and building again gives the same bytes
    low = engine.describe_plan(plan, lowered=True)
    n_igemm = sum(1 for o in low["ops"] if o["kind"] == "conv" and o.get("igemm"))
    x = torch.from_numpy(synth.images(B, S, S, seed=21))
    e1 = engine.Engine(plan)
    try:
        t1 = e1.tactics()
        assert len(t1) == n_igemm and all(r["candidates"] >= 1 and r["tactic"] and r["default"] for r in t1)
        # timed in place when the plan was built: the records carry what the builder measured, and a moved layer was faster
        assert all(r["default_us"] > 0 for r in t1)
        # (layers with identical signatures share one decision - the first one's - so compare the step, not every layer)
        assert sum(r["us"] for r in t1) <= 1.02 * sum(r["default_us"] for r in t1)
        o1 = _run(e1, x, gpu)
        monkeypatch.setenv("TRTX_TUNE", "0")
        e0 = engine.Engine(plan)          # TRTX_TUNE=0: the static defaults, whatever the plan carries
        try:
            # tune_engine returns before recording anything: no tactic is applied, whatever the plan carries ...
            assert e0.tactics() == []
            # ... and the engine then runs exactly what an engine of a plan BUILT under TRTX_TUNE=0 runs (this used to assert nothing)
            plan0 = engine.build_plan("yolov8n", path, batch=B, h=S, w=S, fp16=1, mark_heads=1)
            assert not engine.describe_plan(plan0)["tactics_timed"]
            o0 = _run(e0, x, gpu)
            e00 = engine.Engine(plan0)
            try:
                o00 = _run(e00, x, gpu)
            finally:
                e00.close()
            for k in o0:
                if k != "output":
                    assert torch.equal(o0[k], o00[k]), f"TRTX_TUNE=0 does not override the plan's tactics on '{k}'"
        finally:
            e0.close()
        monkeypatch.delenv("TRTX_TUNE")
        e2 = engine.Engine(plan)          # the plan's own choices again: no timing at deserialize
        try:
            t2 = e2.tactics()
            assert [r["tactic"] for r in t2] == [r["tactic"] for r in t1]
            o2 = _run(e2, x, gpu)
            for k in o1:
                if k == "output":         # decode buffer: [count, count x 90 floats, untouched tail]
                    a, b = o1[k].reshape(B, -1), o2[k].reshape(B, -1)
                    assert torch.equal(a[:, 0], b[:, 0])
                    for i in range(B):
                        m = 1 + int(a[i, 0]) * 90
                        assert torch.equal(a[i, :m], b[i, :m])
                else:
                    assert torch.equal(o1[k], o2[k]), f"two engines of one plan differ on '{k}'"
        finally:
            e2.close()
    finally:
        e1.close()


def test_engines_for_contexts_in_flight_choose_work_efficient_tactics(gpu, monkeypatch):
    monkeypatch.delenv("TRTX_TUNE", raising=False)
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=6, h=352, w=352, fp16=1, aux_streams=0)
    e = engine.Engine(plan)
    try:
        for r in e.tactics():
            if r["tactic"].startswith("igemm") and r["default"].startswith("igemm"):
                rows, cols, _ = (int(v) for v in r["tactic"].split()[1].split("x"))
                _, dcols, _ = (int(v) for v in r["default"].split()[1].split("x"))
                # no 64-row tiles; column tiles: the default or the shared 64-wide one.  ("default" is the baseline the search started
                # from - the static default or the winning whole-network palette.)
                assert rows >= 128 and (cols == dcols or 64 in (cols, dcols)), r
            assert not r["tactic"].startswith("wsk")   # work-efficient sets never split K over the waves
    finally:
        e.close()


def test_tactic_cache_file_spares_a_second_process_the_timing(gpu, tmp_path):
    """TRTX_TACTIC_CACHE (the ITimingCache analogue): the first process times (at build) and writes, the second process reads the file,
    times nothing and builds the SAME plan: same kernels, same recorded measurements, byte for byte."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = tmp_path / "tactics.txt"
    script = (
        "import json, sys, os\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))\n"
        "import torch\n"
        "from tensorrtx_amd import engine\n"
        "from util import synth_wts\n"
        "path, _ = synth_wts('yolov8n')\n"
        "import time, hashlib\n"
        "t0 = time.time()\n"
        "plan = engine.build_plan('yolov8n', path, batch=4, h=224, w=224, fp16=1)\n"
        "dt = time.time() - t0\n"
        "e = engine.Engine(plan)\n"
        "print(json.dumps(dict(tactics=e.tactics(), sha=hashlib.sha256(plan).hexdigest(), build_s=dt)))\n"
        "e.close()\n")
    env = dict(os.environ, TRTX_TACTIC_CACHE=str(cache))
    env.pop("TRTX_TUNE", None)
    runs = []
    for _ in range(2):
        out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    first, second = runs
    assert cache.exists() and len(cache.read_text().splitlines()) >= 10
    assert first["sha"] == second["sha"]
    assert [r["tactic"] for r in first["tactics"]] == [r["tactic"] for r in second["tactics"]]
    assert all(r["default_us"] > 0 for r in first["tactics"])
