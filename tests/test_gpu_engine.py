"""GPU parity of the engine (lowered, fused plan executed by the HIP kernels) against the PyTorch-CPU fp32
restatements of the reference builders, on the same seeded synthetic .wts and inputs.

Tolerances (north_star: 1e-4 logit / 1e-3 box IoU vs the fp32 reference):
  * fp32 engines: 1e-4 absolute on logits/probabilities (accumulation-order differences only);
  * fp16 engines (kFP16: fp16 storage, fp32 accumulate): the measured drift after 60 fp16-rounded layers
    is ~1e-2 on O(10) logits, so logits are checked at 5e-2 absolute and boxes through IoU >= 0.98; the
    1e-4 figure is only reachable in the fp32 build.  Both numbers are asserted below.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import models_torch as mt
from oracle import wts as owts
from oracle import yolo_post as yp
from tensorrtx_amd import capi, engine, synth
from util import synth_wts

pytestmark = pytest.mark.gpu


def _run(plan, inputs, batch, gpu):
    e = engine.Engine(plan)
    bufs = []
    for i in range(e.nb_bindings):
        if e.is_input[i]:
            bufs.append(torch.from_numpy(np.ascontiguousarray(inputs[e.names[i]], dtype=np.float32)).to(gpu))
        else:
            n = int(np.prod(e.dims[i])) * (batch if len(e.dims[i]) and not plan_is_explicit(plan) else 1)
            bufs.append(torch.full((n,), float("nan"), dtype=torch.float32, device=gpu))
    e.enqueue(batch, bufs)
    torch.cuda.synchronize()
    out = {e.names[i]: bufs[i].cpu() for i in range(e.nb_bindings) if not e.is_input[i]}
    e.close()
    return out


def plan_is_explicit(plan):
    return engine.describe_plan(plan)["explicit_batch"]


from tests import parity  # noqa: E402  (the table of bounds; check() records what was measured and asserts)


def test_lenet_fp32(gpu):
    path, _ = synth_wts("lenet")
    plan = engine.build_plan("lenet", path, batch=1)
    x = torch.randn(1, 1, 32, 32, generator=torch.Generator().manual_seed(3))
    out = _run(plan, {"data": x.numpy()}, 1, gpu)["prob"]
    ref = mt.lenet(mt.Params(owts.load_wts(path)), x).reshape(-1)
    err = (out - ref).abs().max().item()
    parity.check("lenet_fp32", max_abs_err=err)


@pytest.mark.parametrize("fp16", [0, 1])
def test_resnet50_small(gpu, fp16):
    path, _ = synth_wts("resnet50")
    plan = engine.build_plan("resnet50", path, batch=4, fp16=fp16, h=64, w=64)
    x = torch.rand(3, 3, 64, 64, generator=torch.Generator().manual_seed(4))  # batch 3 < max_batch 4
    out = _run(plan, {"data": x.numpy()}, 3, gpu)["prob"][:3000].reshape(3, 1000)
    with torch.inference_mode():
        ref = mt.resnet50(mt.Params(owts.load_wts(path)), x)
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    parity.check("resnet50_64", "fp16" if fp16 else "fp32", max_abs_err=err, ref_max=scale)
    assert bool((out.argmax(1) == ref.argmax(1)).all())


def test_resnet50_fp16_224_batch32_all_ones_and_random(gpu):
    """Config 2 at full size.  The reference's own smoke input is an all-ones tensor (resnet50.cpp:327-365)."""
    path, _ = synth_wts("resnet50")
    plan = engine.build_plan("resnet50", path, batch=32, fp16=1, h=224, w=224)
    x = torch.rand(32, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    x[0] = 1.0
    out = _run(plan, {"data": x.numpy()}, 32, gpu)["prob"].reshape(32, 1000)
    with torch.inference_mode():
        ref = mt.resnet50(mt.Params(owts.load_wts(path)), x[:4])
    err = (out[:4] - ref).abs().max().item()
    scale = ref.abs().max().item()
    parity.check("resnet50_224_b32", max_abs_err=err, ref_max=scale)
    assert torch.isfinite(out).all()


def _yolo_case(gpu, fp16, size, batch, seed):
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=batch, h=size, w=size, fp16=fp16, mark_heads=1)
    x = torch.from_numpy(synth.images(batch, size, size, seed=seed))
    out = _run(plan, {"images": x.numpy()}, batch, gpu)
    with torch.inference_mode():
        heads, strides = mt.yolov8_det(mt.Params(owts.load_wts(path)), x)
    return out, heads, strides


def test_yolov8n_fp32_engine_matches_oracle(gpu):
    out, heads, strides = _yolo_case(gpu, 0, 128, 2, 5)
    worst = 0.0
    for i, h in enumerate(heads):
        got = out[f"head{i}"].reshape(h.shape)
        worst = max(worst, (got - h).abs().max().item())
    parity.check("yolov8n_fp32_128", head_max_abs_err=worst)   # logits of O(10): inside the north_star's 1e-4
    dec_ref = yp.decode_c([h.numpy() for h in heads], 80, 128, 128, strides)
    dec = out["output"].reshape(2, -1).numpy()
    assert np.array_equal(dec[:, 0], dec_ref[:, 0])


def test_yolov8n_fp32_engine_640_batch32_on_the_fp32_mfma_meets_the_north_star_tolerance(gpu):
    """The build without kFP16 (yolov8/include/config.h:1-3 USE_FP32; yolov8/src/model.cpp:314-324) at the BENCH configuration - 640 x 640, batch 32,
    production plan: fp32 stem kernel, 62 convolutions on v_mfma_f32_16x16x4_f32, folded upsamples, fused DFL + decode on the fp32 head tensors.
    BASELINE's tolerance is asserted on it: head logits within 1e-4 of the fp32 oracle, every oracle candidate found again with box IoU >= 1 - 1e-3.
    The logits come from a second engine of the same build whose head tensors are marked as outputs (same kernels up to the heads; that plan keeps the
    un-fused detect tail), the boxes from the production plan itself."""
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=32, h=640, w=640, fp16=0)
    low = engine.describe_plan(plan, lowered=True)
    kinds = [o["kind"] for o in low["ops"]]
    assert kinds.count("yolo_head") == 1 and sum(1 for o in low["ops"] if o.get("igemm")) >= 62 and low["ops"][0].get("stem")
    x = torch.from_numpy(synth.images(32, 640, 640, seed=1))
    out = _run(plan, {"images": x.numpy()}, 32, gpu)
    dec = out["output"].reshape(32, -1).numpy()
    nb = 4
    with torch.inference_mode():
        heads, strides = mt.yolov8_det(mt.Params(owts.load_wts(path)), x[:nb])
    dec_ref = yp.decode_c([h.numpy() for h in heads], 80, 640, 640, strides)
    st = _match_detections(dec[:nb], dec_ref)
    assert st["ref"] > 50
    plan_h = engine.build_plan("yolov8n", path, batch=nb, h=640, w=640, fp16=0, mark_heads=1)
    out_h = _run(plan_h, {"images": x[:nb].numpy()}, nb, gpu)
    worst = max((out_h[f"head{i}"].reshape(h.shape) - h).abs().max().item() for i, h in enumerate(heads))
    # ... and against the graph evaluated in DOUBLE: the value both fp32 implementations are roundings of.  The fp32 oracle itself sits 5.4e-5 from it on
    # these images (and moves by 1e-5 with its thread count), so "within 1e-4 of the fp32 oracle" compares two round-offs; the ceiling is stated on this one
    with torch.inference_mode():
        heads64, _ = mt.yolov8_det(mt.Params64(owts.load_wts(path)), x[:nb].double())
    worst64 = max((out_h[f"head{i}"].reshape(h.shape).double() - h).abs().max().item() for i, h in enumerate(heads64))
    oracle64 = max((h.double() - h64).abs().max().item() for h, h64 in zip(heads, heads64))
    parity.check("yolov8n_fp32_640_b32", head_max_abs_err_vs_fp64=worst64, head_max_abs_err=worst, matched_fraction=st["matched"] / st["ref"], min_iou=st["min_iou"],
                 max_conf_err=st["max_conf_err"], fp32_oracle_vs_fp64=oracle64, largest_logit=max(h.abs().max().item() for h in heads64),
                 counts=dec[:nb, 0].tolist(), ref_counts=dec_ref[:, 0].tolist())
    # batch-position invariance at fp32 too: image 5 alone returns what image 5 of the batch returned
    plan1 = engine.build_plan("yolov8n", path, batch=1, h=640, w=640, fp16=0)
    one = _run(plan1, {"images": x[5:6].numpy()}, 1, gpu)["output"].reshape(1, -1).numpy()
    n = int(one[0, 0])
    assert n == int(dec[5, 0])
    assert np.array_equal(one[0, 1:1 + n * 90].reshape(n, 90)[:, :6], dec[5, 1:1 + n * 90].reshape(n, 90)[:, :6])


def _match_detections(dec, dec_ref, conf_margin=0.02):
    """Compare decode buffers as sets keyed by (level-cell order is shared): every reference candidate that is
    not within `conf_margin` of the 0.1 threshold must appear with the same class and IoU ~ 1."""
    stats = dict(ref=0, matched=0, min_iou=1.0, max_conf_err=0.0)
    for b in range(dec_ref.shape[0]):
        nr, ng = int(dec_ref[b, 0]), int(dec[b, 0])
        R = dec_ref[b, 1:1 + nr * 90].reshape(nr, 90)[:, :6]
        G = dec[b, 1:1 + ng * 90].reshape(ng, 90)[:, :6]
        if nr >= 1000 or ng >= 1000:
            continue  # overflow: slot sets differ legitimately when a borderline cell flips
        for r in R:
            if abs(r[4] - 0.1) < conf_margin:
                continue
            stats["ref"] += 1
            # candidates come from distinct cells: match on box centre
            c = np.abs((G[:, 0] + G[:, 2]) - (r[0] + r[2])) + np.abs((G[:, 1] + G[:, 3]) - (r[1] + r[3]))
            same = np.nonzero((G[:, 5] == r[5]))[0]
            if len(same) == 0:
                continue
            j = same[np.argmin(c[same])]
            g = G[j]
            ix = max(0.0, min(r[2], g[2]) - max(r[0], g[0])) * max(0.0, min(r[3], g[3]) - max(r[1], g[1]))
            ua = (r[2] - r[0]) * (r[3] - r[1]) + (g[2] - g[0]) * (g[3] - g[1]) - ix
            iou = ix / ua if ua > 0 else 0.0
            if iou > 0.9:
                stats["matched"] += 1
                stats["min_iou"] = min(stats["min_iou"], float(iou))
                stats["max_conf_err"] = max(stats["max_conf_err"], float(abs(g[4] - r[4])))
    return stats


def test_yolov8n_fp16_engine_640(gpu):
    """Config 3 (fp16, 640x640): head tensors and decoded detections against the fp32 oracle."""
    out, heads, strides = _yolo_case(gpu, 1, 640, 4, 1)
    worst_cls, worst_box = 0.0, 0.0
    for i, h in enumerate(heads):
        got = out[f"head{i}"].reshape(h.shape)
        worst_box = max(worst_box, (got[:, :4] - h[:, :4]).abs().max().item())
        worst_cls = max(worst_cls, (got[:, 4:] - h[:, 4:]).abs().max().item())
    dec_ref = yp.decode_c([h.numpy() for h in heads], 80, 640, 640, strides)
    dec = out["output"].reshape(4, -1).numpy()
    st = _match_detections(dec, dec_ref)
    # fp16 storage of 63 layers vs the fp32 oracle: 0.065-0.077 on O(10) class logits, 0.015-0.02 cells on the DFL distances, min IoU
    # 0.9967-0.9975 (profiles/r0*_parity_metrics.jsonl; bounds in tests/parity.py at <= 1.5x the worst run).  The north_star's 1e-4 / 1e-3
    # bar is met by the fp32 build only (test_yolov8n_fp32_engine_matches_oracle).
    assert st["ref"] > 50
    parity.check("yolov8n_fp16_640", cls_logit_max_abs_err=worst_cls, box_ltrb_max_abs_err=worst_box, counts=dec[:, 0].tolist(),
                 ref_counts=dec_ref[:, 0].tolist(), matched_fraction=st["matched"] / st["ref"], **st)


def test_yolov8n_fp16_fused_head_and_stem_640(gpu):
    """The production plan (no debug outputs): fp32-NCHW stem kernel + fused DFL/decode kernel instead of the
    shuffle/softmax/concat chain.  Checked against the fp32 oracle's decode."""
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=4, h=640, w=640, fp16=1)
    low = engine.describe_plan(plan, lowered=True)
    assert [o["kind"] for o in low["ops"]].count("yolo_head") == 1
    x = torch.from_numpy(synth.images(4, 640, 640, seed=1))
    out = _run(plan, {"images": x.numpy()}, 4, gpu)
    with torch.inference_mode():
        heads, strides = mt.yolov8_det(mt.Params(owts.load_wts(path)), x)
    dec_ref = yp.decode_c([h.numpy() for h in heads], 80, 640, 640, strides)
    dec = out["output"].reshape(4, -1).numpy()
    st = _match_detections(dec, dec_ref)
    assert st["ref"] > 50
    parity.check("yolov8n_fp16_640_fused", counts=dec[:, 0].tolist(), ref_counts=dec_ref[:, 0].tolist(), matched_fraction=st["matched"] / st["ref"], **st)
    assert np.abs(dec[:, 0] - dec_ref[:, 0]).max() <= 0.02 * dec_ref[:, 0].max() + 3


def test_yolov8n_fp16_engine_640_batch32_the_bench_configuration(gpu):
    """Config 3 exactly as bench.py runs it (fp16, 640x640, BATCH 32, production plan).  Full-size checks that do not need 32 oracle
    forward passes: (a) images 0..3 against the fp32 oracle as at batch 4; (b) permuting the images of the batch permutes the
    outputs bit for bit (no cross-image leakage in tiles that span image borders, slot compaction per image); (c) the kept set of
    the device NMS is the oracle NMS of the engine's own decode buffer, for all 32 images."""
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=32, h=640, w=640, fp16=1)
    e = engine.Engine(plan)
    x = torch.from_numpy(synth.images(32, 640, 640, seed=1))
    perm = torch.tensor([(7 * i + 3) % 32 for i in range(32)])
    outs = []
    for xin in (x, x[perm]):
        out = torch.full((32, 1 + 1000 * 90), float("nan"), dtype=torch.float32, device=gpu)
        e.enqueue(32, [xin.to(gpu), out])
        torch.cuda.synchronize()
        outs.append(out)
    dec, dec_p = outs[0].cpu().numpy(), outs[1].cpu().numpy()
    # (b) record j of the permuted run is image perm[j] of the first run; only the written part of each row is defined
    for j in range(32):
        n = int(dec_p[j, 0])
        assert n == int(dec[perm[j], 0]) and 0 < n <= 1000
        assert np.array_equal(dec_p[j, 1:1 + n * 90].reshape(n, 90)[:, :6], dec[perm[j], 1:1 + n * 90].reshape(n, 90)[:, :6])
    # (a)
    with torch.inference_mode():
        heads, strides = mt.yolov8_det(mt.Params(owts.load_wts(path)), x[:4])
    dec_ref = yp.decode_c([h.numpy() for h in heads], 80, 640, 640, strides)
    st = _match_detections(dec[:4], dec_ref)
    assert st["ref"] > 50
    parity.check("yolov8n_fp16_640_b32", counts=dec[:4, 0].tolist(), ref_counts=dec_ref[:, 0].tolist(), matched_fraction=st["matched"] / st["ref"], **st)
    # (c)
    ki, kc, kd = capi.yolo_nms(outs[0])
    torch.cuda.synchronize()
    ri, rc, rd = yp.batch_nms_c(np.nan_to_num(dec))
    assert np.array_equal(kc.cpu().numpy(), rc)
    for b in range(32):
        assert np.array_equal(ki.cpu().numpy()[b, :rc[b]], ri[b, :rc[b]])
    e.close()


def test_engine_decode_then_nms_pipeline(gpu):
    """enqueue -> YoloLayer plugin output stays on the device -> trtx_yolo_nms; the kept boxes must be the
    oracle NMS of the engine's own decode buffer (bit-exact selection on identical inputs)."""
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=8, h=320, w=320, fp16=1)
    e = engine.Engine(plan)
    x = torch.from_numpy(synth.images(8, 320, 320, seed=9)).to(gpu)
    out = torch.empty((8, 1 + 1000 * 90), dtype=torch.float32, device=gpu)
    e.enqueue(8, [x, out])
    ki, kc, kd = capi.yolo_nms(out)
    torch.cuda.synchronize()
    ri, rc, rd = yp.batch_nms_c(out.cpu().numpy())
    assert np.array_equal(kc.cpu().numpy(), rc)
    for b in range(8):
        assert np.array_equal(ki[b, :rc[b]].cpu().numpy(), ri[b, :rc[b]])
    e.close()


def _retina_match(dec, ref, conf_margin=0.01):
    """Match decoded anchors between engine and oracle buffers (both in canonical anchor order): anchors whose
    oracle confidence is not within `conf_margin` of the 0.02 cut must appear with nearly identical boxes."""
    stats = dict(ref=0, matched=0, max_box_err=0.0, max_conf_err=0.0, min_iou=1.0)
    for b in range(ref.shape[0]):
        nr, ng = int(ref[b, 0]), int(dec[b, 0])
        R = ref[b, 1:1 + nr * 15].reshape(nr, 15)
        G = dec[b, 1:1 + ng * 15].reshape(ng, 15)
        j = 0
        for r in R:
            if abs(r[4] - 0.02) < conf_margin:
                continue
            stats["ref"] += 1
            # both lists are ordered by anchor; advance until the landmark-0 x coordinate (an affine function of the
            # anchor position) is close
            best = None
            for jj in range(j, min(j + 8, ng)):
                if np.abs(G[jj, :4] - r[:4]).max() < 2.0:
                    best = jj
                    break
            if best is None:
                continue
            j = best + 1
            stats["matched"] += 1
            stats["max_box_err"] = max(stats["max_box_err"], float(np.abs(G[best, :4] - r[:4]).max()))
            stats["max_conf_err"] = max(stats["max_conf_err"], float(abs(G[best, 4] - r[4])))
            g = G[best]
            ix = max(0.0, min(r[2], g[2]) - max(r[0], g[0])) * max(0.0, min(r[3], g[3]) - max(r[1], g[1]))
            ua = (r[2] - r[0]) * (r[3] - r[1]) + (g[2] - g[0]) * (g[3] - g[1]) - ix
            stats["min_iou"] = min(stats["min_iou"], float(ix / ua) if ua > 0 else 0.0)
    return stats


@pytest.mark.parametrize("hw,batch", [((256, 320), 2), ((1280, 1280), 1)])
def test_retinaface_r50_fp16_engine(gpu, hw, batch):
    """Config 4: R50 body + FPN (deconv-as-upsample) + SSH + heads + Decode_TRT plugin, fp16, vs the fp32 oracle."""
    from oracle import det_post as dp
    H, W = hw
    path, _ = synth_wts("retinaface_r50")
    plan = engine.build_plan("retinaface_r50", path, batch=batch, fp16=1, h=H, w=W)
    x = (torch.from_numpy(synth.images(batch, H, W, seed=2)) * 255 - 110) / 64
    out = _run(plan, {"data": x.numpy()}, batch, gpu)["prob"].reshape(batch, -1).numpy()
    with torch.inference_mode():
        heads = mt.retinaface_r50(mt.Params(owts.load_wts(path)), x)
    ref = dp.retina_decode([h.reshape(batch, 32, -1).numpy() for h in heads], H, W)
    st = _retina_match(out, ref)
    assert st["ref"] > 100
    # boxes are exp()-scaled anchors up to several hundred px wide: judged by IoU (north_star: 1e-3 box IoU is the fp32 budget; fp16
    # storage measures 8e-3 .. 1e-2) and by the absolute error in px; bounds in tests/parity.py
    parity.check("retinaface_r50_fp16", f"{hw[0]}x{hw[1]}", counts=out[:, 0].tolist(), ref_counts=ref[:, 0].tolist(),
                 matched_fraction=st["matched"] / st["ref"], **st)
    # device NMS on the engine's own decode buffer == oracle NMS of the same buffer
    from tensorrtx_amd import det_ops
    gi_, gc_, _ = det_ops.retina_nms(torch.from_numpy(out).to(gpu), H, W)
    ri_, rc_ = dp.retina_nms(out)
    assert np.array_equal(gc_.cpu().numpy(), rc_)
    for b in range(batch):
        assert np.array_equal(gi_.cpu().numpy()[b, :rc_[b]], ri_[b, :rc_[b]])


@pytest.mark.parametrize("cin,cout,fp16", [(16, 8, 1), (24, 16, 1), (64, 32, 1), (16, 8, 0)])
def test_kernel_equals_stride_deconvolution_small_channel_counts(gpu, cin, cout, fp16):
    """a 2x2/2 ConvTranspose is lowered to a 1x1 conv + depth-to-space; with Cin 16 / 24 the stand-in conv
    (K = Cin < 32) does not take the MFMA path and the direct kernel must still see the re-laid-out weights and bias."""
    import torch.nn.functional as F
    from tensorrtx_amd import builder
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(2, cin, 9, 7, generator=g)
    w = torch.randn(cin, cout, 2, 2, generator=g) * 0.2   # CKRS
    b = torch.randn(cout, generator=g)
    net = builder.Network(max_batch=2, fp16=bool(fp16))
    t = net.input("data", (cin, 9, 7))
    l = net.conv(t, w.numpy(), b.numpy(), stride=2, deconv=True)
    net.mark_output(net.out(l), "out")
    plan = net.build()
    net.close()
    got = _run(plan, {"data": x.numpy()}, 2, gpu)["out"].reshape(2, cout, 18, 14)
    ref = F.conv_transpose2d(x.half().float() if fp16 else x, w.half().float() if fp16 else w, b, stride=2)
    assert (got - ref).abs().max().item() < (2e-2 if fp16 else 1e-4)


def test_engine_is_bound_to_its_device_and_device_replicas_run(gpu):
    """Multi-GPU in one process (tutorials/multi_GPU_processing.md:13-30): an engine reports the device it was deserialized on, and
    DeviceReplicas shards a global batch over per-device replicas, each on its own stream (one device on the test box: this
    covers the path, not the scaling).  (LeNet is no use here: the reference graph, lenet.cpp:95-118, multiplies W x flat^T and is
    only meaningful for N = 1.)"""
    from tensorrtx_amd import replicas
    path, _ = synth_wts("resnet50")
    plan = engine.build_plan("resnet50", path, batch=4, fp16=0, h=64, w=64)
    reps = replicas.DeviceReplicas([torch.cuda.current_device()], lambda d: engine.Engine(plan))
    try:
        e = reps.engines[0]
        assert e.device == torch.cuda.current_device()
        shards = reps.shards(4)
        assert [len(s) for s in shards] == [4]
        x = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(5))
        xin = x.to(gpu)
        out = torch.empty(4 * 1000, dtype=torch.float32, device=gpu)
        torch.cuda.synchronize()
        reps.enqueue([4], [[xin if e.is_input[i] else out for i in range(e.nb_bindings)]])
        reps.synchronize()
        with torch.inference_mode():
            ref = mt.resnet50(mt.Params(owts.load_wts(path)), x)
        assert (out.cpu().reshape(4, 1000) - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    finally:
        reps.close()


def test_yolov8n_fp16_folded_upsample_is_bit_identical_to_the_resize_launches(gpu):
    """Upsample -> Concat -> Conv1x1 folded into the convolution's A-gather (lower.cpp fold_upsample, ConvArgs::up_in) against the same plan
    lowered with the two nearest-resize launches (TRTX_FOLD_UPSAMPLE=0).  With every layer on a plain implicit-GEMM tile (those sum K in
    one order; the weight-stationary and wave-split-K kernels, which the unfolded 1x1s may otherwise take, are switched off for the
    process) the same operands are multiplied in the same order: every output float is the same.  Own process: the switches are read once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = f"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import numpy as np, torch
from tensorrtx_amd import engine, synth
from util import synth_wts
from test_gpu_engine import _run
path, _ = synth_wts('yolov8n')
B = 5
plan = engine.build_plan('yolov8n', path, batch=B, h=640, w=640, fp16=1, mark_heads=1)
x = synth.images(B, 640, 640, seed=21)
outs = []
for fold in ('1', '0'):
    os.environ['TRTX_FOLD_UPSAMPLE'] = fold
    low = engine.describe_plan(plan, lowered=True)
    assert [o['kind'] for o in low['ops']].count('resize') == (0 if fold == '1' else 2)
    outs.append(_run(plan, {{'images': x}}, B, torch.device('cuda:0')))
for k in outs[0]:
    a, b = outs[0][k].view(torch.int32), outs[1][k].view(torch.int32)   # bit patterns: unused decode slots keep their NaN fill
    assert torch.equal(a, b), (k, int((a != b).sum()))
assert outs[0]['output'].reshape(B, -1)[:, 0].min() > 0
print('IDENTICAL')
"""
    env = dict(os.environ, TRTX_TUNE="0", TRTX_CONV_NOWS="1", TRTX_CONV_NOWSK="1")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("fp16", [True, False])
def test_folded_upsample_small_network_matches_torch(gpu, monkeypatch, fp16):
    """The Upsample -> Concat -> Conv1x1 pattern on an ad-hoc network (ragged map 16 x 24 -> 8 x 12, batch below max_batch): folded and unfolded
    lowering against a plain PyTorch fp32 evaluation of the same layers - in an fp16 engine and (round 5) in an fp32 engine, whose folded and
    unfolded forms must agree to fp32 round-off with the reference."""
    import torch.nn.functional as F
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_runtime_cpu import _upsample_concat_net
    plan, w = _upsample_concat_net(fp16=fp16)
    x = torch.randn(2, 3, 16, 24, generator=torch.Generator().manual_seed(1))
    t = {k: torch.from_numpy(v) for k, v in w.items()}
    low = F.relu(F.conv2d(x, t["a"], stride=2, padding=1))
    skip = F.relu(F.conv2d(x, t["b"], stride=1, padding=1))
    ref = F.relu(F.conv2d(torch.cat([F.interpolate(low, scale_factor=2, mode="nearest"), skip], 1), t["c"]))
    scale = ref.abs().max().item()
    for fold in ("1", "0"):
        monkeypatch.setenv("TRTX_FOLD_UPSAMPLE", fold)
        kinds = [o["kind"] for o in engine.describe_plan(plan, lowered=True)["ops"]]
        assert ("resize" in kinds) == (fold == "0")
        got = _run(plan, {"data": x.numpy()}, 2, gpu)["y"].reshape(ref.shape)
        err = (got - ref).abs().max().item()
        assert err < (4e-3 if fp16 else 1e-5) * max(scale, 1.0), (fold, err, scale)


@pytest.mark.parametrize("fp16,fused", [(0, True), (1, True), (0, False), (1, False)])
def test_conv_bn_mish_engine_matches_the_reference_expression(gpu, fp16, fused):
    """g1: Conv -> Scale(BN) -> Mish_TRT (convBnMish, yolov4/yolov4.cpp:199-213) as ONE launch with the Mish epilogue, against the
    reference plugin's expression (oracle/mish.py = yolov4/mish.cu:111-135) applied to a PyTorch fp32 convolution.  fp32: 1e-4 (the
    north_star logit figure; the sum order of the convolution is the only difference).  fp16: inputs and weights rounded to fp16 in the
    oracle too, so what is left is the fp16 rounding of the OUTPUT (half an ulp of |y| <= 16: 4e-3) plus the summation order."""
    import torch.nn.functional as F
    from oracle import mish as om
    from test_runtime_cpu import _conv_bn_mish_net
    plan, (w, scale, shift) = _conv_bn_mish_net(bool(fp16), fused=fused)
    x = torch.randn(2, 16, 12, 20, generator=torch.Generator().manual_seed(9))
    got = _run(plan, {"data": x.numpy()}, 2, gpu)["out"]
    wt = torch.from_numpy(w * scale[:, None, None, None])           # the engine folds the Scale into the weights before rounding them
    xin = x.half().float() if fp16 else x
    if fp16:
        wt = wt.half().float()
    y = F.conv2d(xin, wt, torch.from_numpy(shift), padding=1)
    if not fused:
        y = F.max_pool2d(y.half().float() if fp16 else y, 2, 2)
    ref = om.mish_torch(y)
    got = got.reshape(ref.shape)
    err = (got - ref).abs().max().item()
    assert err < (8e-3 if fp16 else 1e-4), err
    assert (ref.abs() > 1).any() and (ref < 0).any()   # both tails of the activation are exercised


@pytest.mark.parametrize("tuned", [False, True], ids=["static", "tuned"])
def test_grouped_sibling_convolutions_return_the_bits_of_one_launch_each(gpu, tuned):
    """lower.cpp group_convs / conv_igemm_group_f16_kernel: the detect head's 18 convolutions in 6 launches of 3 sibling layers each - every
    member computed exactly as its own launch computes it.  Both engines on the static kernel choice, same input: every head tensor and the
    decode buffer equal bit for bit; and the grouped plan really has the 6 groups.  No TRTX_CONV_NOWSK since round 5: a group member
    is marked "never wave-split-K", so its own launch - TRTX_GROUP_CONVS=0 here, or the executor's per-member fallback at another batch - walks K in
    the grouped kernel's order even on the 20 x 20 level, where the static rule would otherwise split K over the waves.  (Own process: the switch
    is read while the plan is lowered, and engines cache per-process tactic choices.)
    "tuned" (round 6): the same with the tactics TIMED at build, in a latency engine (whose candidate sets hold the wave-split-K and weight-stationary
    kernels - another K order): the would-be members carry ConvArgs::k_pinned, conv_tactics() lists only main-kernel-order candidates for them, and every other
    layer has one signature in both plans, i.e. one choice per process."""
    import subprocess
    code = r'''
import sys, numpy as np, torch
TUNED = %d
sys.path.insert(0, "tests")
from tensorrtx_amd import engine, synth
from util import synth_wts
import os
path, _ = synth_wts("yolov8n")
B, S = 8, 640
x = torch.from_numpy(synth.images(B, S, S, seed=31)).cuda()
outs = []
for grp in ("1", "0"):
    os.environ["TRTX_GROUP_CONVS"] = grp
    plan = engine.build_plan("yolov8n", path, batch=B, h=S, w=S, fp16=1, mark_heads=1, **({} if TUNED else {"aux_streams": 0}))
    kinds = [o["kind"] for o in engine.describe_plan(plan, lowered=True)["ops"]]
    assert kinds.count("conv_group") == (6 if grp == "1" else 0), kinds
    e = engine.Engine(plan)
    bufs = [x] + [torch.zeros(B * int(np.prod(e.dims[i])), dtype=torch.float32, device="cuda") for i in range(1, e.nb_bindings)]
    e.enqueue(B, bufs)
    torch.cuda.synchronize()
    outs.append({e.names[i]: bufs[i].cpu() for i in range(1, e.nb_bindings)})
    e.close()
for k in outs[0]:
    a, b = outs[0][k], outs[1][k]
    if k == "output":
        a, b = a.reshape(B, -1), b.reshape(B, -1)
        assert torch.equal(a[:, 0], b[:, 0])
        for i in range(B):
            m = 1 + int(a[i, 0]) * 90
            assert torch.equal(a[i, :m], b[i, :m]), k
    else:
        assert torch.equal(a, b), k
print("GROUPED_EQUALS_SINGLE", sorted(outs[0]))
''' % (1 if tuned else 0)
    env = dict(os.environ)
    env.pop("TRTX_TUNE", None)
    env.pop("TRTX_TACTIC_CACHE", None)
    if not tuned:
        env["TRTX_TUNE"] = "0"
    env.pop("TRTX_CONV_NOWSK", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "GROUPED_EQUALS_SINGLE" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("cin,cout,k,stride,hw", [(3, 64, 7, 2, (45, 161)), (3, 64, 7, 2, (37, 130)), (3, 16, 3, 2, (33, 67)), (1, 32, 3, 1, (20, 135)), (3, 64, 7, 2, (40, 128))])
def test_stem_kernel_on_ragged_widths(gpu, cin, cout, k, stride, hw):
    """conv_stem_lds_kernel (round 4: persistent, unconditional weight loads, widths that are not a multiple of 4 - Faster R-CNN's 1333 -
    through the 16-byte DMA rows + a scalar fix-up of the chunk that straddles the row end): a first layer on an fp32 NCHW input against
    PyTorch on fp16-rounded operands (the kernel feeds fp16 MFMA operands, fp32 accumulate; one fp16 rounding of the output)."""
    import torch.nn.functional as F
    from tensorrtx_amd import builder
    g = torch.Generator().manual_seed(cin * 1000 + cout + hw[1])
    x = torch.rand(3, cin, *hw, generator=g) * 2 - 1
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    b = torch.randn(cout, generator=g) * 0.1
    net = builder.Network(max_batch=3, fp16=True)
    t = net.input("data", (cin,) + hw)
    l = net.conv(t, w.numpy(), b.numpy(), stride=stride, padding=k // 2)
    net.mark_output(net.out(net.activation(net.out(l), "relu")), "out")
    plan = net.build()
    net.close()
    low = engine.describe_plan(plan, lowered=True)
    assert [o for o in low["ops"] if o["kind"] == "conv"][0]["stem"]
    ref = F.relu(F.conv2d(x.half().float(), w.half().float(), b, stride=stride, padding=k // 2))
    got = _run(plan, {"data": x.numpy()}, 3, gpu)["out"].reshape(ref.shape)
    err = (got - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), err      # half an fp16 ulp of the output + the summation order
    assert (got[..., -1] - ref[..., -1]).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())   # the last output column reads the ragged tail
