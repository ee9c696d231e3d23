"""CPU tests of the RetinaFace / R-CNN oracle restatements (oracle/csrc/retina_rcnn_ref.c) against independent
NumPy computations and hand-made cases.  Reference: retinaface/decode.cu:110-165, common.hpp:91-130,
rcnn/RpnDecode.cu, RpnNms.cu, RoiAlign.cu, PredictorDecode.cu, BatchedNms.cu."""
import numpy as np

from oracle import det_post as dp
from tensorrtx_amd import synth

FLT_MAX = np.finfo(np.float32).max


def test_retina_decode_against_numpy_formula():
    H, W = 64, 96
    ins = synth.retina_head_tensors(2, H, W, faces=20, seed=1)
    out = dp.retina_decode(ins, H, W)
    for b in range(2):
        recs = []
        for l, s in enumerate((8, 16, 32)):
            h, w = H // s, W // s
            x = ins[l][b].astype(np.float64)
            anchor = 16 * 4 ** l
            for idx in range(h * w):
                for k in range(2):
                    c1, c2 = ins[l][b, 8 + 2 * k, idx], ins[l][b, 8 + 2 * k + 1, idx]
                    conf = np.float32(np.exp(c2, dtype=np.float32) / (np.exp(c1, dtype=np.float32) + np.exp(c2, dtype=np.float32)))
                    if conf <= 0.02:
                        continue
                    p0, p1 = np.float32((idx % w + 0.5) / w), np.float32((idx // w + 0.5) / h)
                    p2, p3 = np.float32(anchor * (k + 1) / W), np.float32(anchor * (k + 1) / H)
                    cx = p0 + x[4 * k + 0, idx] * 0.1 * p2
                    cy = p1 + x[4 * k + 1, idx] * 0.1 * p3
                    bw = p2 * np.exp(x[4 * k + 2, idx] * 0.2)
                    bh = p3 * np.exp(x[4 * k + 3, idx] * 0.2)
                    recs.append([(cx - bw / 2) * W, (cy - bh / 2) * H, (cx + bw / 2) * W, (cy + bh / 2) * H, conf,
                                 (p0 + x[12 + 10 * k, idx] * 0.1 * p2) * W])
        recs = np.asarray(recs)
        n = int(out[b, 0])
        assert n == len(recs) and n > 20
        got = out[b, 1:1 + n * 15].reshape(n, 15)
        assert np.allclose(got[:, :5], recs[:, :5], rtol=1e-5, atol=1e-4)
        assert np.allclose(got[:, 5], recs[:, 5], rtol=1e-5, atol=1e-4)


def test_retina_nms_hand_case():
    out = np.zeros((1, 1 + 6 * 15), np.float32)

    def put(i, box, conf):
        out[0, 1 + 15 * i:1 + 15 * i + 5] = [*box, conf]

    put(0, [0, 0, 10, 10], 0.9)
    put(1, [1, 1, 11, 11], 0.8)      # iou 0.68 -> suppressed
    put(2, [20, 20, 30, 30], 0.1)    # float 0.1f > double 0.1 -> kept by "<= 0.1" test
    put(3, [40, 40, 50, 50], 0.05)   # below threshold
    put(4, [20, 20, 30, 31], 0.95)   # suppresses #2
    out[0, 0] = 5
    idx, cnt = dp.retina_nms(out)
    assert list(idx[0, :cnt[0]]) == [4, 0]


def test_rpn_decode_top_n_and_empty_boxes():
    anchors = dp.generate_anchors()
    assert anchors.shape == (60,) and np.allclose(anchors[4:8], [-16, -16, 16, 16])
    s, d = synth.rcnn_rpn_tensors(1, 15, 6, 7, seed=2)
    os_, ob = dp.rpn_decode(s.reshape(1, -1), d.reshape(1, -1), 6, 7, 96, 112, 16.0, anchors, 50)
    order = np.argsort(-s.reshape(-1), kind="stable")[:50]
    valid = os_[0] > -FLT_MAX
    assert np.array_equal(os_[0][valid], s.reshape(-1)[order][valid])
    assert (ob[0, :, 0] >= 0).all() and (ob[0, :, 2] <= 112).all() and (ob[0, :, 3] <= 96).all()
    # fewer scores than top_n: identity order, tail = -FLT_MAX
    os2, _ = dp.rpn_decode(s.reshape(1, -1), d.reshape(1, -1), 6, 7, 96, 112, 16.0, anchors, 1000)
    assert (os2[0, 630:] == -FLT_MAX).all()


def test_rpn_nms_and_batched_nms_small():
    boxes = np.array([[[0, 0, 10, 10], [1, 1, 11, 11], [30, 30, 40, 40], [0, 0, 10, 10.5]]], np.float32)
    scores = np.array([[0.9, 0.95, 0.5, 0.2]], np.float32)
    out = dp.rpn_nms(scores, boxes, 3, 0.5)
    # order: #1 (0.95) kept, #0 suppressed, #2 kept, #3 suppressed; re-sort keeps suppressed after survivors
    assert np.array_equal(out[0], boxes[0][[1, 2, 0]])
    cls = np.array([[1, 1, 1, 2]], np.float32)
    os_, ob, oc = dp.batched_nms(0, scores, boxes, cls, 4, 0.5)
    assert np.allclose(os_[0], [0.95, 0.5, 0.2, 0.0]) and list(oc[0]) == [1, 1, 2, 1]
    os1, _, _ = dp.batched_nms(1, scores, boxes, cls, 4, 0.5)   # soft-linear: 0.9 * (1 - iou)
    iou = (9 * 9) / (100 + 100 - 81)
    assert np.isclose(os1[0, 1], 0.5) and np.isclose(sorted(os1[0])[1], 0.9 * (1 - iou), atol=1e-6)
    os2, _, _ = dp.batched_nms(2, scores, boxes, cls, 4, 0.5)   # soft-gaussian
    assert np.isclose(sorted(os2[0])[1], 0.9 * np.exp(-iou * iou / 0.5), atol=1e-6)


def test_roi_align_constant_and_linear_maps():
    feats = np.zeros((1, 2, 8, 10), np.float32)
    feats[0, 0] = 3.0
    feats[0, 1] = np.arange(10, dtype=np.float32)[None, :]  # f(y, x) = x: bilinear sampling is exact in the interior
    boxes = np.array([[[32, 32, 96, 96], [16, 16, 48, 80]]], np.float32)
    out = dp.roi_align(boxes, feats, 2, 1 / 16.0)
    assert np.allclose(out[0, :, 0], 3.0)
    # roi 0: x in [1.5, 5.5] (aligned: *1/16 - 0.5); bin centres over a 2x2 grid per bin -> mean x = 2.5, 4.5
    assert np.allclose(out[0, 0, 1], [[2.5, 4.5], [2.5, 4.5]], atol=1e-5)


def test_predictor_decode_small():
    s, d, p = synth.rcnn_box_head_tensors(1, n=20, classes=5, seed=3)
    os_, ob, oc = dp.predictor_decode(s, d, p, 800, 1333)
    order = np.argsort(-s.reshape(-1), kind="stable")[:20]
    assert np.array_equal(oc[0], (order % 5).astype(np.float32))
    ok = os_[0] > 0
    assert np.array_equal(os_[0][ok], s.reshape(-1)[order][ok])
    i = order[0]
    n, c = i // 5, i % 5
    w, h = p[0, n, 2] - p[0, n, 0], p[0, n, 3] - p[0, n, 1]
    cx = d[0, n, c, 0] / 10 * w + p[0, n, 0] + 0.5 * w
    pw = np.exp(d[0, n, c, 2] / 5) * w
    assert np.isclose(ob[0, 0, 0], max(0.0, cx - 0.5 * pw), rtol=1e-5, atol=1e-3)
    assert (ob[0, :, 3] <= 1333).all()  # reference clips y2 with image_width
