"""GPU parity of the RetinaFace and R-CNN plugin operators (C ABI section 1) against the sequential C oracle.
Index / selection outputs are bit-exact on identical inputs; values that go through expf are compared at 1e-5
relative (device expf vs glibc expf), everything else bit-exact."""
import numpy as np
import pytest

from oracle import det_post as dp
from tensorrtx_amd import det_ops, synth

pytestmark = pytest.mark.gpu
FLT_MAX = np.finfo(np.float32).max


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("hw,batch", [((64, 96), 3), ((480, 640), 2), ((1280, 1280), 1)])
def test_retina_decode(gpu, hw, batch):
    H, W = hw
    ins = synth.retina_head_tensors(batch, H, W, faces=200, seed=H)
    ref = dp.retina_decode(ins, H, W)
    got = det_ops.retina_decode([_t(x, gpu) for x in ins], H, W).cpu().numpy()
    assert np.array_equal(got[:, 0], ref[:, 0])
    for b in range(batch):
        n = int(ref[b, 0])
        g = got[b, 1:1 + n * 15].reshape(n, 15)
        r = ref[b, 1:1 + n * 15].reshape(n, 15)
        assert np.allclose(g, r, rtol=2e-6, atol=2e-4)       # px coordinates up to ~1e3: 1 ulp ~ 6e-5
        assert np.array_equal(g[:, 5:], r[:, 5:])            # landmarks: no transcendental, bit exact
    assert ref[:, 0].min() > 100


@pytest.mark.parametrize("hw,batch", [((64, 96), 3), ((480, 640), 2), ((1280, 1280), 1)])
def test_retina_nms_bit_exact(gpu, hw, batch):
    H, W = hw
    ins = synth.retina_head_tensors(batch, H, W, faces=200, seed=7 + H)
    dec = dp.retina_decode(ins, H, W)  # identical input for both sides
    ri, rc = dp.retina_nms(dec)
    gi, gc, gd = det_ops.retina_nms(_t(dec, gpu), H, W)
    gi, gc, gd = gi.cpu().numpy(), gc.cpu().numpy(), gd.cpu().numpy()
    assert np.array_equal(gc, rc)
    for b in range(batch):
        assert np.array_equal(gi[b, :rc[b]], ri[b, :rc[b]])
        rec = dec[b, 1:].reshape(-1, 15)[ri[b, :rc[b]]]
        assert np.array_equal(gd[b, :rc[b]], rec)
    assert rc.min() > 20


def test_retina_nms_empty_and_ties(gpu):
    H, W = 64, 96
    n_f = dp.retina_out_floats(H, W)
    dec = np.zeros((2, n_f), np.float32)
    rec = dec[1, 1:].reshape(-1, 15)
    rec[:40, :4] = [10, 10, 20, 20]
    rec[:40, 4] = 0.7          # 40 identical boxes, identical conf: slot order decides, first one survives
    rec[40:60, 0] = np.arange(20) * 30
    rec[40:60, 1] = 0
    rec[40:60, 2] = np.arange(20) * 30 + 20
    rec[40:60, 3] = 20
    rec[40:60, 4] = 0.7
    dec[1, 0] = 60
    ri, rc = dp.retina_nms(dec)
    gi, gc, _ = det_ops.retina_nms(_t(dec, gpu), H, W)
    assert np.array_equal(gc.cpu().numpy(), rc) and rc[0] == 0
    assert np.array_equal(gi.cpu().numpy()[1, :rc[1]], ri[1, :rc[1]])


@pytest.mark.parametrize("batch,h,w,top_n", [(4, 50, 84, 6000), (2, 13, 9, 100), (1, 6, 7, 1000)])
def test_rpn_decode(gpu, batch, h, w, top_n):
    anchors = dp.generate_anchors()
    s, d = synth.rcnn_rpn_tensors(batch, 15, h, w, seed=h)
    s[:, :, 0, 0] = s[:, :, 1, 1]  # exact score ties: the stable sort keeps the lower index first
    rs, rb = dp.rpn_decode(s.reshape(batch, -1), d.reshape(batch, -1), h, w, h * 16, w * 16, 16.0, anchors, top_n)
    gs, gb = det_ops.rpn_decode(_t(s.reshape(batch, -1), gpu), _t(d.reshape(batch, -1), gpu), h, w, h * 16, w * 16, 16.0, anchors, top_n)
    gs, gb = gs.cpu().numpy(), gb.cpu().numpy()
    assert np.array_equal(gs, rs)                          # selection + empty-box marking identical
    assert np.allclose(gb, rb, rtol=2e-6, atol=1e-3)       # expf on the box sizes


def test_rpn_nms_bit_exact_full_size(gpu):
    """6000 -> 1000 at IoU 0.7 (rcnn.cpp:37-40) on boxes produced by the decode oracle."""
    anchors = dp.generate_anchors()
    s, d = synth.rcnn_rpn_tensors(4, 15, 50, 84, seed=3)
    rs, rb = dp.rpn_decode(s.reshape(4, -1), d.reshape(4, -1), 50, 84, 800, 1333, 16.0, anchors, 6000)
    ref = dp.rpn_nms(rs, rb, 1000, 0.7)
    got = det_ops.rpn_nms(_t(rs, gpu), _t(rb, gpu), 1000, 0.7).cpu().numpy()
    assert np.array_equal(got, ref)


def test_rpn_nms_fewer_survivors_than_post(gpu):
    rng = np.random.default_rng(0)
    boxes = np.tile(np.array([[0, 0, 50, 50]], np.float32), (300, 1)) + rng.uniform(0, 3, size=(300, 4)).astype(np.float32)
    boxes[150:] += 200
    scores = rng.uniform(0, 1, size=(1, 300)).astype(np.float32)
    scores[0, 5] = -FLT_MAX  # an empty proposal from RpnDecode
    ref = dp.rpn_nms(scores, boxes[None], 64, 0.7)
    got = det_ops.rpn_nms(_t(scores, gpu), _t(boxes[None], gpu), 64, 0.7).cpu().numpy()
    assert np.array_equal(got, ref)


def test_roi_align(gpu):
    rng = np.random.default_rng(1)
    feats = rng.normal(size=(2, 32, 50, 84)).astype(np.float32)
    x1 = rng.uniform(-20, 1300, size=(2, 64)); y1 = rng.uniform(-20, 780, size=(2, 64))
    boxes = np.stack([x1, y1, x1 + rng.uniform(1, 500, size=(2, 64)), y1 + rng.uniform(1, 400, size=(2, 64))], -1).astype(np.float32)
    ref = dp.roi_align(boxes, feats, 14, 1 / 16.0, 0)
    got = det_ops.roi_align(_t(boxes, gpu), _t(feats, gpu), 14, 1 / 16.0, 0).cpu().numpy()
    assert np.array_equal(got, ref)  # only IEEE basic ops (no FMA contraction on either side)
    ref2 = dp.roi_align(boxes, feats, 7, 1 / 16.0, 2)
    got2 = det_ops.roi_align(_t(boxes, gpu), _t(feats, gpu), 7, 1 / 16.0, 2).cpu().numpy()
    assert np.array_equal(got2, ref2)


def test_roi_align_full_size_and_degenerate_boxes(gpu):
    """C5 geometry (1000 proposals on a 50x84 map, 64 of the 1024 channels) + boxes outside / larger than the image, zero-area
    and inverted boxes (count = 0 -> NaN, as the reference computes 0/0)."""
    rng = np.random.default_rng(11)
    feats = rng.normal(size=(1, 64, 50, 84)).astype(np.float32)
    x1 = rng.uniform(0, 1300, size=(1, 1000)); y1 = rng.uniform(0, 780, size=(1, 1000))
    boxes = np.stack([x1, y1, np.minimum(x1 + rng.uniform(2, 900, size=(1, 1000)), 1333), np.minimum(y1 + rng.uniform(2, 700, size=(1, 1000)), 800)], -1)
    boxes = boxes.astype(np.float32)
    boxes[0, 0] = [-500, -500, 2500, 2500]     # far larger than the map: 13x13 samples per bin, most of them outside
    boxes[0, 1] = [100, 100, 100, 100]         # zero area: grid 0 -> 0/0
    boxes[0, 2] = [300, 300, 200, 250]         # inverted
    boxes[0, 3] = [-4000, 10, -3000, 60]       # entirely left of the map: every sample invalid -> 0
    ref = dp.roi_align(boxes, feats, 14, 1 / 16.0, 0)
    got = det_ops.roi_align(_t(boxes, gpu), _t(feats, gpu), 14, 1 / 16.0, 0).cpu().numpy()
    assert np.array_equal(got, ref, equal_nan=True)
    assert np.isnan(ref[0, 1]).all() and (ref[0, 3] == 0).all()


def test_roi_align_native_nhwc_f16(gpu):
    """The engine-native form (NHWC fp16 in / out) against the fp32 oracle on the fp16-rounded feature map."""
    import torch
    rng = np.random.default_rng(12)
    feats = rng.normal(size=(2, 128, 25, 42)).astype(np.float16)
    x1 = rng.uniform(-20, 600, size=(2, 40)); y1 = rng.uniform(-20, 360, size=(2, 40))
    boxes = np.stack([x1, y1, x1 + rng.uniform(1, 400, size=(2, 40)), y1 + rng.uniform(1, 300, size=(2, 40))], -1).astype(np.float32)
    ref = dp.roi_align(boxes, feats.astype(np.float32), 14, 1 / 16.0, 0)              # [B, P, C, 14, 14]
    nhwc = torch.from_numpy(np.ascontiguousarray(feats.transpose(0, 2, 3, 1))).to(gpu)
    got = det_ops.roi_align_nhwc_f16(_t(boxes, gpu), nhwc, 14, 1 / 16.0, 0).float().cpu().numpy()  # [B*P, 14, 14, C]
    got = got.reshape(2, 40, 14, 14, 128).transpose(0, 1, 4, 2, 3)
    assert np.abs(got - ref).max() < 4e-3   # one fp16 rounding of O(1) values
    ref2 = dp.roi_align(boxes, feats.astype(np.float32), 7, 1 / 16.0, 2)
    got2 = det_ops.roi_align_nhwc_f16(_t(boxes, gpu), nhwc, 7, 1 / 16.0, 2).float().cpu().numpy().reshape(2, 40, 7, 7, 128).transpose(0, 1, 4, 2, 3)
    assert np.abs(got2 - ref2).max() < 4e-3


def test_predictor_decode(gpu):
    s, d, p = synth.rcnn_box_head_tensors(4, 1000, 80, seed=5)
    rs, rb, rc = dp.predictor_decode(s, d, p, 800, 1333)
    gs, gb, gc = det_ops.predictor_decode(_t(s, gpu), _t(d, gpu), _t(p, gpu), 800, 1333)
    assert np.array_equal(gs.cpu().numpy(), rs) and np.array_equal(gc.cpu().numpy(), rc)
    assert np.allclose(gb.cpu().numpy(), rb, rtol=2e-6, atol=1e-3)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_batched_nms(gpu, method):
    s, d, p = synth.rcnn_box_head_tensors(4, 1000, 80, seed=6)
    ps, pb, pc = dp.predictor_decode(s, d, p, 800, 1333)  # identical inputs for both sides
    rs, rb, rc = dp.batched_nms(method, ps, pb, pc, 100, 0.5)
    gs, gb, gc = det_ops.batched_nms(method, _t(ps, gpu), _t(pb, gpu), _t(pc, gpu), 100, 0.5)
    gs, gb, gc = gs.cpu().numpy(), gb.cpu().numpy(), gc.cpu().numpy()
    if method == 2:  # expf in the gaussian decay: selection may only differ where scores are within 1 ulp
        assert np.allclose(gs, rs, rtol=1e-5, atol=1e-7)
        same = (gc == rc).mean()
        assert same > 0.97
    else:
        assert np.array_equal(gs, rs) and np.array_equal(gb, rb) and np.array_equal(gc, rc)


def test_rcnn_chain_end_to_end(gpu):
    """RpnDecode -> RpnNms -> RoiAlign on the device, each stage fed by the previous stage's device output."""
    anchors = dp.generate_anchors()
    s, d = synth.rcnn_rpn_tensors(2, 15, 50, 84, seed=9)
    feats = np.random.default_rng(2).normal(size=(2, 16, 50, 84)).astype(np.float32)
    gs, gb = det_ops.rpn_decode(_t(s.reshape(2, -1), gpu), _t(d.reshape(2, -1), gpu), 50, 84, 800, 1333, 16.0, anchors, 6000)
    props = det_ops.rpn_nms(gs, gb, 1000, 0.7)
    roi = det_ops.roi_align(props, _t(feats, gpu), 14, 1 / 16.0, 0)
    # oracle chain on the GPU's own intermediate outputs (isolates each stage)
    ref_props = dp.rpn_nms(gs.cpu().numpy(), gb.cpu().numpy(), 1000, 0.7)
    assert np.array_equal(props.cpu().numpy(), ref_props)
    ref_roi = dp.roi_align(ref_props[:, :50], feats, 14, 1 / 16.0, 0)
    assert np.array_equal(roi.cpu().numpy()[:, :50], ref_roi)
