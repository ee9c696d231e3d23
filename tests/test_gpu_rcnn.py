"""GPU parity of the Faster R-CNN R50-C4 engine (BASELINE config 5) against the oracle.

The detector is a chain of discrete decisions (top-k, two NMS passes), so fp16 noise upstream can legitimately flip a
choice and change everything downstream.  fp32 is compared end to end; fp16 is compared stage by stage, restarting the
oracle from the tensors the engine itself produced ("features", "proposals" taps) so each stage is judged on its own.
"""
import functools

import numpy as np
import pytest
import torch

from oracle import models_torch as mt
from oracle import wts as owts
from tensorrtx_amd import engine, synth
from test_gpu_engine import _run
from tests import parity
from util import synth_wts

pytestmark = pytest.mark.gpu


def _images(batch, H, W, seed):
    return torch.from_numpy(synth.images(batch, H, W, seed=seed)).permute(0, 2, 3, 1).contiguous() * 255


def _iou_matrix(a, b):
    x0 = np.maximum(a[:, None, 0], b[None, :, 0])
    y0 = np.maximum(a[:, None, 1], b[None, :, 1])
    x1 = np.minimum(a[:, None, 2], b[None, :, 2])
    y1 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x1 - x0, 0, None) * np.clip(y1 - y0, 0, None)
    ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None, :] - inter
    return np.where(ua > 0, inter / np.maximum(ua, 1e-12), 0.0)


def _matched_fraction(got, ref, thr=0.9):
    """fraction of non-empty reference boxes that have a partner with IoU > thr among `got`"""
    ref = ref[(ref[:, 2] > ref[:, 0]) & (ref[:, 3] > ref[:, 1])]
    if len(ref) == 0:
        return 1.0
    return float((_iou_matrix(ref, got).max(1) > thr).mean())


def test_rcnn_fp32_engine_end_to_end(gpu):
    cfg = dict(pre_nms_topk=300, post_nms_topk=50, detections=20)
    path, _ = synth_wts("rcnn_r50c4")
    B, H, W = 2, 128, 160
    plan = engine.build_plan("rcnn_r50c4", path, batch=B, fp16=0, h=H, w=W, mark_stages=1, **cfg)
    x = _images(B, H, W, 7)
    out = _run(plan, {"images": x.numpy()}, B, gpu)
    with torch.inference_mode():
        ref = mt.rcnn_r50c4(mt.Params(owts.load_wts(path)), x, pre_nms_topk=300, post_nms_topk=50, detections_per_image=20)
    feats = out["features"].reshape(ref["features"].shape).numpy()
    err = float(np.abs(feats - ref["features"].numpy()).max())
    props = out["proposals"].reshape(B, 50, 4).numpy()
    scores = out["scores"].reshape(B, 20).numpy()
    boxes = out["boxes"].reshape(B, 20, 4).numpy()
    labels = out["labels"].reshape(B, 20).numpy()
    pm = min(_matched_fraction(props[b], ref["proposals"][b], 0.99) for b in range(B))
    dm = min(_matched_fraction(boxes[b], ref["boxes"][b], 0.95) for b in range(B))
    parity.check("rcnn_fp32", feat_err=err, proposals_matched=pm, detections_matched=dm, score_err=float(np.abs(scores - ref["scores"]).max()))
    assert np.array_equal(labels, ref["labels"])


@pytest.mark.parametrize("hw,batch,cfg", [((320, 416), 2, dict(pre_nms_topk=2000, post_nms_topk=200, detections=50)),
                                           ((800, 1067), 1, dict()),
                                           ((800, 1333), 4, dict())])  # BASELINE configs[4]: 1333x800, batch 4
def test_rcnn_fp16_engine_stagewise(gpu, hw, batch, cfg):
    H, W = hw
    path, _ = synth_wts("rcnn_r50c4")
    plan = engine.build_plan("rcnn_r50c4", path, batch=batch, fp16=1, h=H, w=W, mark_stages=1, **cfg)
    x = _images(batch, H, W, 11)
    out = _run(plan, {"images": x.numpy()}, batch, gpu)
    ocfg = dict(pre_nms_topk=cfg.get("pre_nms_topk", 6000), post_nms_topk=cfg.get("post_nms_topk", 1000),
                detections_per_image=cfg.get("detections", 100))
    P, D = ocfg["post_nms_topk"], ocfg["detections_per_image"]
    params = mt.Params(owts.load_wts(path))
    fdim = lambda n: functools.reduce(lambda v, _: (v - 1) // 2 + 1, range(4), n)  # noqa: E731  four stride-2 stages
    fh, fw = fdim(H), fdim(W)
    feats = out["features"].reshape(batch, 1024, fh, fw)
    props = out["proposals"].reshape(batch, P, 4).numpy()
    with torch.inference_mode():
        # stage 1: backbone, fp16 storage vs fp32
        full = mt.rcnn_r50c4(params, x, stage="backbone", **ocfg)
        rf = full["features"]
        rel = float((feats - rf).abs().max() / rf.abs().max())
        feats4 = feats
        # stage 2: RPN + decode + NMS restarted from the engine's own features
        s2 = mt.rcnn_r50c4(params, x, given={"features": feats4}, **ocfg)
        pm = min(_matched_fraction(props[b], s2["proposals"][b], 0.9) for b in range(batch))
        # stage 3: RoIAlign + res5 + predictor + soft-NMS restarted from the engine's own features and proposals
        s3 = mt.rcnn_r50c4(params, x, given={"features": feats4, "proposals": props}, **ocfg)
    scores = out["scores"].reshape(batch, D).numpy()
    boxes = out["boxes"].reshape(batch, D, 4).numpy()
    labels = out["labels"].reshape(batch, D).numpy()
    dm = min(_matched_fraction(boxes[b], s3["boxes"][b], 0.85) for b in range(batch))
    top = float(np.abs(scores[:, 0] - s3["scores"][:, 0]).max())
    assert np.isfinite(scores).all() and scores[:, 0].min() > 0.05
    # measured over runs: backbone 1.4e-3 .. 1.9e-3 relative, proposals matched 99.3-100 %, detections 97-100 % (tests/parity.py)
    parity.check("rcnn_fp16", f"{hw[0]}x{hw[1]}", feat_rel_err=rel, proposals_matched=pm, detections_matched=dm, top_score_err=top,
                 labels_equal=float((labels == s3["labels"]).mean()))


def test_mask_rcnn_fp32_engine(gpu):
    """Mask R-CNN head on the GPU (fp32, small config): masks of the engine's own detections vs the oracle restarted from
    the engine's features / boxes / labels; and the plugin alone against its NumPy restatement."""
    from oracle import det_post as dp
    from tensorrtx_amd import det_ops
    cfg = dict(pre_nms_topk=300, post_nms_topk=50, detections=20)
    path, _ = synth_wts("rcnn_r50c4")
    B, H, W = 2, 128, 160
    plan = engine.build_plan("rcnn_r50c4", path, batch=B, fp16=0, h=H, w=W, mark_stages=1, mask=1, **cfg)
    x = _images(B, H, W, 7)
    out = _run(plan, {"images": x.numpy()}, B, gpu)
    fdim = lambda n: functools.reduce(lambda v, _: (v - 1) // 2 + 1, range(4), n)  # noqa: E731
    feats = out["features"].reshape(B, 1024, fdim(H), fdim(W))
    boxes = out["boxes"].reshape(B, 20, 4).numpy()
    labels = out["labels"].reshape(B, 20).numpy()
    with torch.inference_mode():
        ref = mt.rcnn_r50c4(mt.Params(owts.load_wts(path)), x, pre_nms_topk=300, post_nms_topk=50, detections_per_image=20,
                            mask_on=True, given={"features": feats, "proposals": out["proposals"].reshape(B, 50, 4).numpy(),
                                                 "boxes": boxes, "labels": labels})
    masks = out["masks"].reshape(B, 20, 1, 14, 14).numpy()
    err = float(np.abs(masks - ref["masks"]).max())
    parity.check("mask_rcnn_fp32", mask_err=err, mask_mean=float(masks.mean()))
    assert masks.std() > 0.01
    # the plugin operator by itself
    g = torch.Generator().manual_seed(3)
    lab = torch.randint(-1, 6, (2, 7), generator=g).float()
    m = torch.randn(2, 7, 5, 6, 6, generator=g)
    got = det_ops.mask_rcnn_inference(lab.to(gpu), m.to(gpu)).cpu().numpy()
    assert np.allclose(got, dp.mask_select(lab.numpy(), m.numpy()), atol=1e-6)


def test_mask_rcnn_fp16_engine(gpu):
    """fp16 build: the 2x2/2 ConvTranspose runs as a 1x1 MFMA implicit GEMM + depth-to-space; masks vs the oracle restarted
    from the engine's own features / boxes / labels."""
    cfg = dict(pre_nms_topk=500, post_nms_topk=100, detections=30)
    path, _ = synth_wts("rcnn_r50c4")
    B, H, W = 2, 160, 224
    plan = engine.build_plan("rcnn_r50c4", path, batch=B, fp16=1, h=H, w=W, mark_stages=1, mask=1, **cfg)
    kinds = [o["kind"] for o in engine.describe_plan(plan, lowered=True)["ops"]]
    assert "depth_to_space" in kinds and "deconv" not in kinds
    x = _images(B, H, W, 9)
    out = _run(plan, {"images": x.numpy()}, B, gpu)
    fdim = lambda n: functools.reduce(lambda v, _: (v - 1) // 2 + 1, range(4), n)  # noqa: E731
    feats = out["features"].reshape(B, 1024, fdim(H), fdim(W))
    boxes = out["boxes"].reshape(B, 30, 4).numpy()
    labels = out["labels"].reshape(B, 30).numpy()
    with torch.inference_mode():
        ref = mt.rcnn_r50c4(mt.Params(owts.load_wts(path)), x, pre_nms_topk=500, post_nms_topk=100, detections_per_image=30,
                            mask_on=True, given={"features": feats, "proposals": out["proposals"].reshape(B, 100, 4).numpy(),
                                                 "boxes": boxes, "labels": labels})
    masks = out["masks"].reshape(B, 30, 1, 14, 14).numpy()
    err = float(np.abs(masks - ref["masks"]).max())
    parity.check("mask_rcnn_fp16", mask_err=err, mask_mean=float(masks.mean()))
    assert np.isfinite(masks).all() and masks.std() > 0.01


def test_roi_align_stride_fold_is_bit_identical(gpu, monkeypatch):
    """The 7x7 RoIAlign + stride-1 readers against the 14x14 RoIAlign + stride-2 readers: the same samples summed in the same order, the same
    GEMM rows - every output of the engine equal bit for bit (Faster R-CNN, fp16, two images)."""
    path, _ = synth_wts("rcnn_r50c4")
    cfg = dict(pre_nms_topk=2000, post_nms_topk=200, detections=50)
    x = _images(2, 320, 416, 23)
    monkeypatch.setenv("TRTX_TUNE", "0")      # one static kernel choice for both plans (a 7x7-pixel GEMM and a strided one get different tactics otherwise)
    folded = _run(engine.build_plan("rcnn_r50c4", path, batch=2, fp16=1, h=320, w=416, **cfg), {"images": x.numpy()}, 2, gpu)
    monkeypatch.setenv("TRTX_ROIALIGN_FOLD_STRIDE", "0")
    full = _run(engine.build_plan("rcnn_r50c4", path, batch=2, fp16=1, h=320, w=416, **cfg), {"images": x.numpy()}, 2, gpu)
    assert set(folded) == set(full)
    for k in folded:
        assert torch.equal(folded[k], full[k]), k
