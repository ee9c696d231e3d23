"""Graph pin: the reference's network builders, compiled unmodified against include/NvInfer.h, must serialize the SAME plan as the product's
host builders (tensorrtx_amd/host/*.cpp) -- byte for byte -- on the same seeded weights.  See tests/ref_builder_cases.py.

Two legs per case, both CPU:
  * live: run oracle/_ref/libref_build_<family>.so (skipped where it could not be built) and compare the bytes;
  * committed: the product plan's SHA-256 equals the digest of the reference builder's plan recorded by
    tests/golden/make_ref_builder_golden.py (holds wherever the seeded weights reproduce, e.g. on the GPU box).
"""
import hashlib
import os
import sys

import pytest

from tests import ref_builder_cases as rb
from tensorrtx_amd import engine


def _first_difference(a: bytes, b: bytes) -> str:
    da, db = engine.describe_plan(a), engine.describe_plan(b)
    if len(da["layers"]) != len(db["layers"]):
        return f"{len(da['layers'])} layers (reference) vs {len(db['layers'])} (product)"
    for i, (x, y) in enumerate(zip(da["layers"], db["layers"])):
        if x != y:
            keys = [k for k in x if x[k] != y.get(k)]
            return f"layer {i} {x.get('name')}: fields {keys}: reference {[x[k] for k in keys]} vs product {[y.get(k) for k in keys]}"
    for i, (x, y) in enumerate(zip(da["tensors"], db["tensors"])):
        if x != y:
            return f"tensor {i}: reference {x} vs product {y}"
    n = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
    return f"same layers and tensors; first differing byte at {n} (weights)"


@pytest.mark.parametrize("case", list(rb.CASES))
def test_reference_builder_and_host_builder_serialize_the_same_plan(case):
    if not os.path.exists(rb.lib_path(case)):
        if os.path.isdir("/root/reference"):
            pytest.fail(f"{rb.lib_path(case)} missing although /root/reference is present: run python oracle/ref_build.py")
        pytest.skip("oracle/_ref builder library not present (cannot be built without /root/reference); the committed digest still pins the plan")
    ref, mine = rb.reference_plan(case), rb.product_plan(case)
    assert ref == mine, _first_difference(ref, mine)


@pytest.mark.parametrize("case", list(rb.CASES))
def test_host_builder_plan_matches_committed_reference_builder_digest(case):
    gold = rb.load_golden()[case]
    if rb.wts_sha(case) != gold["wts_sha256"]:
        pytest.skip("the seeded synthetic weights differ from the ones the digest was recorded on (different torch build)")
    plan = rb.product_plan(case)
    assert len(plan) == gold["plan_bytes"]
    assert hashlib.sha256(plan).hexdigest() == gold["plan_sha256"]


@pytest.mark.gpu
def test_a_plan_serialized_by_the_reference_builder_runs_on_the_gpu(gpu, monkeypatch):
    """The plan the REFERENCE'S buildEngineYolov8Det serialized through the shim (here, next to a GPU, with TRTX_TUNE=0 so that no timing
    noise enters either build) is the host builder's plan byte for byte, deserializes, and detects what the fp32 oracle detects."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from oracle import models_torch as mt
    from oracle import wts as owts
    from oracle import yolo_post as yp
    from tensorrtx_amd import synth
    from test_gpu_engine import _match_detections, _run
    from util import synth_wts
    if not os.path.exists(rb.lib_path("yolov8n_det")):
        pytest.skip("oracle/_ref builder library not present")
    monkeypatch.setenv("TRTX_TUNE", "0")
    ref, mine = rb.reference_plan("yolov8n_det"), rb.product_plan("yolov8n_det")
    assert ref == mine
    x = synth.images(1, 640, 640, seed=1)
    out = _run(ref, {"images": x}, 1, gpu)["output"].reshape(1, -1).numpy()
    path, _ = synth_wts("yolov8n")
    with torch.inference_mode():
        heads, strides = mt.yolov8_det(mt.Params(owts.load_wts(path)), torch.from_numpy(x))
    dec_ref = yp.decode_c([h.numpy() for h in heads], 80, 640, 640, strides)
    st = _match_detections(out, dec_ref)
    assert st["ref"] > 20 and st["matched"] >= 0.99 * st["ref"] and st["min_iou"] > 0.99
