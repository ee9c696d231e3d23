"""YOLOv8 seg / pose / obb ENGINES: the host builder's task graphs (buildEngineYolov8Seg / Pose / Obb of the
reference, yolov8/src/model.cpp:1057-1308, 1310-1563, 2499-2740) through build -> lower -> enqueue, against the torch twin of
the graph and the C oracle of the branch decode (yolov8/plugin/yololayer.cu:222-279)."""
import numpy as np
import pytest
import torch

from oracle import models_torch as mt
from oracle import wts as owts
from oracle import yolo_post as yp
from tensorrtx_amd import engine, synth
from util import synth_wts

pytestmark = pytest.mark.gpu

TASKS = {  # name -> (task id of the host builder, classes, oracle kwargs, weight kwargs)
    "seg": (1, 80, dict(seg=True), dict(cls_bias=-5.0)),
    "pose": (2, 1, dict(pose=True), dict(cls_bias=-1.0)),   # one class: lift the class bias so that cells pass the 0.1 gate
    "obb": (3, 15, dict(obb=True), dict(cls_bias=-4.0)),
}


def _run(plan, x, batch, gpu):
    e = engine.Engine(plan)
    bufs = []
    for i in range(e.nb_bindings):
        if e.is_input[i]:
            bufs.append(x.to(gpu))
        else:
            bufs.append(torch.full((batch * int(np.prod(e.dims[i])),), float("nan"), dtype=torch.float32, device=gpu))
    e.enqueue(batch, bufs)
    torch.cuda.synchronize()
    out = {e.names[i]: bufs[i].cpu() for i in range(e.nb_bindings) if not e.is_input[i]}
    e.close()
    return out


def _case(task, fp16, gpu, size=128, batch=2, seed=7):
    tid, nc, okw, wkw = TASKS[task]
    path, _ = synth_wts(f"yolov8n_{task}", num_class=nc, **wkw)
    plan = engine.build_plan("yolov8n", path, batch=batch, h=size, w=size, fp16=fp16, task=tid, classes=nc, mark_heads=1)
    x = torch.from_numpy(synth.images(batch, size, size, seed=seed))
    out = _run(plan, x, batch, gpu)
    with torch.inference_mode():
        ref = mt.yolov8_det(mt.Params(owts.load_wts(path)), x, num_class=nc, task=task)
    return out, ref, nc, okw


@pytest.mark.parametrize("task", sorted(TASKS))
def test_task_engine_fp32_matches_oracle(gpu, task):
    out, ref, nc, okw = _case(task, 0, gpu)
    heads, strides = ref[0], ref[1]
    worst = 0.0
    for i, h in enumerate(heads):
        worst = max(worst, (out[f"head{i}"].reshape(h.shape) - h).abs().max().item())
    assert worst < 1e-3, worst
    # the plugin ran on the ENGINE's heads: decode those with the C oracle -> bit-exact records in the canonical order
    got_heads = [out[f"head{i}"].reshape(h.shape).numpy() for i, h in enumerate(heads)]
    dec_ref = yp.decode_ex_c(got_heads, nc, 128, 128, strides, kpt_conf=0.0, **okw)
    dec = out["output"].reshape(dec_ref.shape).numpy()
    assert dec_ref[:, 0].min() >= 5, "the synthetic weights must produce candidates for the branch to be exercised"
    assert np.array_equal(dec[:, 0], dec_ref[:, 0])
    # Detection = bbox[4], conf, class_id, mask[32], keypoints[51], angle (yolov8/include/types.h:4-12); the plugin writes the common
    # six floats plus its task's field and leaves the others as they were (the reference only clears the counter, yololayer.cu:285-288).
    # Tolerances are those of the plugin-level test (test_gpu_yolo8_branches.py): device expf vs libm differ in the last bit.
    for b in range(dec.shape[0]):
        n = int(dec_ref[b, 0])
        G, R = dec[b, 1:1 + n * 90].reshape(n, 90), dec_ref[b, 1:1 + n * 90].reshape(n, 90)
        assert np.array_equal(G[:, 5], R[:, 5])
        assert np.allclose(G[:, 4], R[:, 4], rtol=0, atol=2e-7)
        if task == "obb":
            assert np.allclose(G[:, :4], R[:, :4], rtol=1e-6, atol=1e-4) and np.allclose(G[:, 89], R[:, 89], rtol=1e-6, atol=1e-7)
        else:
            assert np.array_equal(G[:, :4], R[:, :4])
        if task == "seg":
            assert np.array_equal(G[:, 6:38], R[:, 6:38])
        if task == "pose":
            assert np.array_equal(G[:, 38:89] == -1, R[:, 38:89] == -1) and np.allclose(G[:, 38:89], R[:, 38:89], rtol=1e-6, atol=1e-4)
    if task == "seg":
        proto = ref[2]
        err = (out["proto"].reshape(proto.shape) - proto).abs().max().item()
        assert err < 1e-3 * max(1.0, proto.abs().max().item()), err


@pytest.mark.parametrize("task", sorted(TASKS))
def test_task_engine_fp16_close_to_oracle(gpu, task):
    out, ref, nc, okw = _case(task, 1, gpu)
    heads, strides = ref[0], ref[1]
    for i, h in enumerate(heads):
        got = out[f"head{i}"].reshape(h.shape)
        assert torch.isfinite(got).all()
        assert (got - h).abs().max().item() < 0.05 * max(1.0, h.abs().max().item())  # fp16 storage between ~25 fused layers
    dec_ref = yp.decode_ex_c([h.numpy() for h in heads], nc, 128, 128, strides, kpt_conf=0.0, **okw)
    dec = out["output"].reshape(dec_ref.shape).numpy()
    assert np.all(np.abs(dec[:, 0] - dec_ref[:, 0]) <= np.maximum(3, 0.15 * dec_ref[:, 0]))  # cells near the 0.1 gate may flip
    if task == "seg":
        proto = ref[2]
        assert (out["proto"].reshape(proto.shape) - proto).abs().max().item() < 0.05 * max(1.0, proto.abs().max().item())
