"""Seeded cases shared by the reference-pinning tests and the golden generator (tests/golden/make_ref_golden.py).

Each case names a reference plugin (family library of oracle/_ref, plugin type), how to construct it (a serialized blob
in the REFERENCE's byte layout, or creator fields), its inputs, and three evaluators of the same inputs:

  ref(case, dev)      the reference's own kernel, run on the MI355X through the C-ABI plugin v-table (oracle/ref.py)
  product(case, dev)  this repo's HIP path through the C ABI (tensorrtx_amd)
  oracle(case)        the CPU restatement under oracle/

All three return a list of numpy arrays in a CANONICAL form (``canon``): the reference's decode plugins append with
atomicAdd, so record order is arbitrary there — records are sorted lexicographically before comparison.
"""
import struct

import numpy as np

from oracle import det_post as dp
from oracle import yolo_post as yp
from tensorrtx_amd import synth

FLT_MAX = np.finfo(np.float32).max


def _sorted_rows(a):
    a = np.ascontiguousarray(a)
    if a.shape[0] == 0:
        return a
    return a[np.lexsort(a.T[::-1])]


def canon_records(out, rec_floats, keep_floats):
    """[B, 1 + n*rec] decode buffers -> per image (count, sorted records[:, :keep])."""
    res = []
    for b in range(out.shape[0]):
        n = int(out[b, 0])
        rec = out[b, 1:1 + n * rec_floats].reshape(n, rec_floats)[:, :keep_floats]
        res.append(_sorted_rows(rec))
    return res


class Case:
    def __init__(self, name, family, plugin, batch, inputs, out_shapes, blob=None, fields=None, canon=None, ref_exact=True,
                 rtol=0.0, atol=0.0, per_image=False):
        self.name, self.family, self.plugin, self.batch = name, family, plugin, batch
        self.inputs, self.out_shapes, self.blob, self.fields = inputs, out_shapes, blob, fields
        self.canon = canon or (lambda outs: [np.asarray(o) for o in outs])
        self.ref_exact = ref_exact          # False: the reference kernel itself races (documented per case)
        self.rtol, self.atol = rtol, atol   # tolerance for values that pass through expf (device vs glibc: 1 ulp)
        self.per_image = per_image          # run the reference one image per enqueue (see rpn_nms_case)


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------------------------------------------ yolov8
def yolov8_decode_case(batch=2, seed=11, size=640):
    ins = synth.yolo_head_tensors(batch, net_h=size, net_w=size, seed=seed)
    info = np.array([80, 17, 0, size, size, 1000, 0, 0, 0, 8, 16, 32], dtype=np.int32)  # block.cpp:267-293 "combinedInfo"
    c = Case(f"yolov8_decode_b{batch}_{size}", "yolov8_plugin", "YoloLayer_TRT", batch, ins, [(batch, 1 + 1000 * 90)],
             fields=[("combinedInfo", info)], canon=lambda outs: canon_records(np.asarray(outs[0]), 90, 6), atol=2e-7)
    c.size = size
    return c


def yolov8_product(c, dev):
    from tensorrtx_amd import capi
    return [capi.yolo_decode([_t(x, dev) for x in c.inputs], 80, c.size, c.size, [8, 16, 32], 1000).cpu().numpy()]


def yolov8_oracle(c):
    return [yp.decode_c(c.inputs, 80, c.size, c.size, [8, 16, 32])]


# ------------------------------------------------------------------------------------------------------ yolov5 (anchor based)
YOLO5_KERNEL = np.dtype([("width", "<i4"), ("height", "<i4"), ("anchors", "<f4", 6)])  # YoloKernel, yolov5/src/types.h:5-9


def yolov5_decode_case(batch=2, seed=12, size=640, seg=False):
    ins = synth.yolov5_head_tensors(batch, net_h=size, net_w=size, seed=seed, seg=seg)
    grids = [(size // s, size // s) for s in (8, 16, 32)]
    kern = np.zeros(3, dtype=YOLO5_KERNEL)
    for i, (gw, gh) in enumerate(grids):
        kern[i] = (gw, gh, synth.YOLOV5_ANCHORS[i])
    netinfo = np.array([80, size, size, 1000, 1 if seg else 0], dtype=np.int32)  # yolov5/src/model.cpp:250-256
    keepf = 38 if seg else 6
    c = Case(f"yolov5_decode_b{batch}_{size}" + ("_seg" if seg else ""), "yolov5_plugin", "YoloLayer_TRT", batch, ins, [(batch, 1 + 1000 * 38)],
             fields=[("netinfo", netinfo), ("kernels", kern.view(np.uint8), 3)],
             canon=lambda outs: canon_records(np.asarray(outs[0]), 38, keepf), rtol=2e-6, atol=2e-6)
    c.p = (size, grids, seg)
    return c


def yolov5_product(c, dev):
    from tensorrtx_amd import capi
    size, grids, seg = c.p
    return [capi.yolov5_decode([_t(x, dev) for x in c.inputs], 80, size, size, grids, synth.YOLOV5_ANCHORS, 1000, seg).cpu().numpy()]


def yolov5_oracle(c):
    size, grids, seg = c.p
    return [yp.v5_decode_c(c.inputs, 80, size, size, grids, synth.YOLOV5_ANCHORS, 1000, seg)]


# ------------------------------------------------------------------------------------------------------ retinaface
def retina_decode_case(batch=1, seed=13):
    H, W = 480, 640  # compile-time INPUT_H / INPUT_W of the reference plugin (retinaface/decode.h:16-17)
    ins = synth.retina_head_tensors(batch, H, W, faces=120, seed=seed)
    c = Case("retina_decode_480x640", "retinaface_plugin", "Decode_TRT", batch, ins, [(batch, dp.retina_out_floats(H, W))], fields=[],
             canon=lambda outs: canon_records(np.asarray(outs[0]), 15, 15), rtol=2e-6, atol=2e-4)
    c.hw = (H, W)
    return c


def retina_product(c, dev):
    from tensorrtx_amd import det_ops
    return [det_ops.retina_decode([_t(x, dev) for x in c.inputs], *c.hw).cpu().numpy()]


def retina_oracle(c):
    return [dp.retina_decode(c.inputs, *c.hw)]


# ------------------------------------------------------------------------------------------------------ rcnn
def rpn_decode_case(batch=2, h=13, w=9, top_n=100, seed=3):
    anchors = dp.generate_anchors()
    s, d = synth.rcnn_rpn_tensors(batch, 15, h, w, seed=seed)
    blob = struct.pack("<iQ", top_n, anchors.size) + anchors.tobytes() + struct.pack("<fQQQQ", 16.0, h, w, h * 16, w * 16)
    c = Case(f"rpn_decode_{h}x{w}_top{top_n}", "rcnn_plugins", "RpnDecode", batch, [s, d], [(batch, top_n, 1), (batch, top_n, 4)],
             blob=blob, rtol=2e-6, atol=1e-3)
    c.p = (h, w, top_n, anchors)
    return c


def rpn_decode_product(c, dev):
    from tensorrtx_amd import det_ops
    h, w, top_n, anchors = c.p
    B = c.batch
    s, b = det_ops.rpn_decode(_t(c.inputs[0].reshape(B, -1), dev), _t(c.inputs[1].reshape(B, -1), dev), h, w, h * 16, w * 16, 16.0, anchors,
                              top_n)
    return [s.cpu().numpy().reshape(B, top_n, 1), b.cpu().numpy()]


def rpn_decode_oracle(c):
    h, w, top_n, anchors = c.p
    B = c.batch
    s, b = dp.rpn_decode(c.inputs[0].reshape(B, -1), c.inputs[1].reshape(B, -1), h, w, h * 16, w * 16, 16.0, anchors, top_n)
    return [s.reshape(B, top_n, 1), b]


def rpn_nms_case(batch=2, pre=600, post=64, seed=5):
    """pre <= 1024: the reference launches ceil(pre/1024) blocks that synchronise with a block-level barrier only
    (RpnNms.cu:93-110), so it is exact greedy NMS for one block and racy beyond.
    per_image: found by this pin — rpnNms() fills its `indices` iota ONCE before the batch loop (RpnNms.cu:84-88) and the
    re-sort of image 0 then overwrites it (RpnNms.cu:112-113), so for images >= 1 of a batch the first sort pairs scores
    with image 0's final permutation and the gathered boxes are wrong.  The reference only ever runs BATCH_SIZE = 1
    (rcnn.cpp:24); the product implements the batch-0 semantics for every image, and the reference is driven one image
    per enqueue here."""
    anchors = dp.generate_anchors()
    h, w = 20, 30
    s, d = synth.rcnn_rpn_tensors(batch, 15, h, w, seed=seed)
    rs, rb = dp.rpn_decode(s.reshape(batch, -1), d.reshape(batch, -1), h, w, h * 16, w * 16, 16.0, anchors, pre)
    blob = struct.pack("<fiQ", 0.7, post, pre)
    c = Case(f"rpn_nms_{pre}_{post}", "rcnn_plugins", "RpnNms", batch, [rs.reshape(batch, pre, 1), rb], [(batch, post, 4)], blob=blob,
             ref_exact=pre <= 1024, per_image=True)
    c.p = (pre, post)
    return c


def rpn_nms_product(c, dev):
    from tensorrtx_amd import det_ops
    pre, post = c.p
    return [det_ops.rpn_nms(_t(c.inputs[0].reshape(c.batch, pre), dev), _t(c.inputs[1], dev), post, 0.7).cpu().numpy()]


def rpn_nms_oracle(c):
    pre, post = c.p
    return [dp.rpn_nms(c.inputs[0].reshape(c.batch, pre), c.inputs[1], post, 0.7)]


def roi_align_case(batch=2, P=12, C=4, fh=25, fw=42, res=14, ratio=0, seed=1):
    rng = np.random.default_rng(seed)
    feats = rng.normal(size=(batch, C, fh, fw)).astype(np.float32)
    x1 = rng.uniform(-20, fw * 16 - 40, size=(batch, P)); y1 = rng.uniform(-20, fh * 16 - 20, size=(batch, P))
    boxes = np.stack([x1, y1, x1 + rng.uniform(1, 400, size=(batch, P)), y1 + rng.uniform(1, 300, size=(batch, P))], -1).astype(np.float32)
    blob = struct.pack("<ifiiiii", res, 1 / 16.0, ratio, P, C, fh, fw)
    c = Case(f"roi_align_P{P}_C{C}_{fh}x{fw}_r{res}_s{ratio}", "rcnn_plugins", "RoiAlign", batch, [boxes, feats], [(batch, P, C, res, res)],
             blob=blob)
    c.p = (res, ratio)
    return c


def roi_align_product(c, dev):
    from tensorrtx_amd import det_ops
    res, ratio = c.p
    return [det_ops.roi_align(_t(c.inputs[0], dev), _t(c.inputs[1], dev), res, 1 / 16.0, ratio).cpu().numpy()]


def roi_align_oracle(c):
    res, ratio = c.p
    return [dp.roi_align(c.inputs[0], c.inputs[1], res, 1 / 16.0, ratio)]


def predictor_decode_case(batch=2, n=200, classes=80, seed=5):
    s, d, p = synth.rcnn_box_head_tensors(batch, n, classes, seed=seed)
    blob = struct.pack("<IIIIQffff", n, classes, 800, 1333, 4, 10.0, 10.0, 5.0, 5.0)
    c = Case(f"predictor_decode_{n}x{classes}", "rcnn_plugins", "PredictorDecode", batch,
             [s.reshape(batch, n, classes, 1, 1), d.reshape(batch, n, classes * 4, 1, 1), p],
             [(batch, n, 1), (batch, n, 4), (batch, n, 1)], blob=blob, rtol=2e-6, atol=1e-3)
    c.p = (n, classes, s, d, p)
    return c


def predictor_decode_product(c, dev):
    from tensorrtx_amd import det_ops
    n, classes, s, d, p = c.p
    a, b, cl = det_ops.predictor_decode(_t(s, dev), _t(d, dev), _t(p, dev), 800, 1333)
    return [a.cpu().numpy().reshape(c.batch, n, 1), b.cpu().numpy(), cl.cpu().numpy().reshape(c.batch, n, 1)]


def predictor_decode_oracle(c):
    n, classes, s, d, p = c.p
    a, b, cl = dp.predictor_decode(s, d, p, 800, 1333)
    return [a.reshape(c.batch, n, 1), b, cl.reshape(c.batch, n, 1)]


def batched_nms_case(method, batch=2, count=200, dets=50, seed=6, class_mod=0):
    """per_image: batchedNms() has the same `indices` reuse across its batch loop as rpnNms() (BatchedNms.cu:115-119 vs
    :146-148, see rpn_nms_case).  class_mod folds the 80 classes onto a few so that the class-aware NMS suppresses a lot."""
    s, d, p = synth.rcnn_box_head_tensors(batch, count, 80, seed=seed)
    ps, pb, pc = dp.predictor_decode(s, d, p, 800, 1333)
    if class_mod:
        pc = np.mod(pc, class_mod).astype(np.float32)
    blob = struct.pack("<ifiQ", method, 0.5, dets, count)
    c = Case(f"batched_nms_m{method}_{count}_{dets}_mod{class_mod}", "rcnn_plugins", "BatchedNms", batch,
             [ps.reshape(batch, count, 1), pb, pc.reshape(batch, count, 1)], [(batch, dets, 1), (batch, dets, 4), (batch, dets, 1)], blob=blob,
             rtol=1e-5 if method == 2 else 0.0, atol=1e-7 if method == 2 else 0.0, per_image=True)
    c.p = (method, count, dets)
    return c


def batched_nms_product(c, dev):
    from tensorrtx_amd import det_ops
    method, count, dets = c.p
    B = c.batch
    a, b, cl = det_ops.batched_nms(method, _t(c.inputs[0].reshape(B, count), dev), _t(c.inputs[1], dev), _t(c.inputs[2].reshape(B, count), dev),
                                   dets, 0.5)
    return [a.cpu().numpy().reshape(B, dets, 1), b.cpu().numpy(), cl.cpu().numpy().reshape(B, dets, 1)]


def batched_nms_oracle(c):
    method, count, dets = c.p
    B = c.batch
    a, b, cl = dp.batched_nms(method, c.inputs[0].reshape(B, count), c.inputs[1], c.inputs[2].reshape(B, count), dets, 0.5)
    return [a.reshape(B, dets, 1), b, cl.reshape(B, dets, 1)]


def mask_case(batch=2, D=10, C=5, S=28, seed=8):
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, C, size=(batch, D, 1)).astype(np.float32)
    masks = rng.normal(0, 2, size=(batch, D, C, S, S)).astype(np.float32)
    c = Case(f"mask_rcnn_inference_D{D}_C{C}_S{S}", "rcnn_plugins", "MaskRcnnInference", batch, [labels, masks], [(batch, D, 1, S, S)],
             blob=struct.pack("<iii", D, S, C), rtol=2e-6, atol=2e-7)
    return c


def mask_product(c, dev):
    from tensorrtx_amd import det_ops
    return [det_ops.mask_rcnn_inference(_t(c.inputs[0].reshape(c.batch, -1), dev), _t(c.inputs[1], dev)).cpu().numpy()]


def mask_oracle(c):
    return [dp.mask_select(c.inputs[0].reshape(c.batch, -1), c.inputs[1])]


# ------------------------------------------------------------------------------------------------------ yolov4 Mish_TRT
def mish_case(batch=2, shape=(5, 7, 9), seed=21):
    """values across every branch of softplus_kernel (yolov4/mish.cu:113-117): |x| <= 30 with the thresholds +-20 and 0 planted"""
    rng = np.random.default_rng(seed)
    n = int(np.prod(shape))
    x = rng.uniform(-30, 30, size=(batch, n)).astype(np.float32)
    x[:, :10] = np.array([0.0, -0.0, 20.0, -20.0, np.nextafter(np.float32(20), np.float32(21)), np.nextafter(np.float32(-20), np.float32(-21)),
                          1e-6, -1e-6, 19.999, -19.999], dtype=np.float32)
    x[:, 10:n // 2] = rng.normal(0, 2, size=(batch, n // 2 - 10)).astype(np.float32)   # where activations actually live
    return Case(f"mish_{'x'.join(map(str, shape))}", "yolov4_plugin", "Mish_TRT", batch, [x.reshape((batch,) + shape)], [(batch,) + shape],
                blob=struct.pack("<i", n), rtol=2e-6, atol=2e-6)   # logf(expf(x) + 1) for x < 0 amplifies one ulp of expf (device vs NumPy) to ~1e-6 absolute


def mish_product(c, dev):
    """the built-in Mish_TRT (plugins/builtin_plugins.cpp -> trtx_mish) through the same v-table route as the reference's"""
    from oracle import ref
    v = ref.make_plugin(ref.registry_get("Mish_TRT"), blob=c.blob)
    return [o.cpu().numpy() for o in ref.run_plugin(v, c.batch, [_t(x, dev) for x in c.inputs], c.out_shapes)]


def mish_oracle(c):
    from oracle import mish as om
    return [om.mish(c.inputs[0])]


def all_cases():
    """(case, product evaluator, oracle evaluator) — small enough that the reference outputs are committed as goldens."""
    return [
        (yolov8_decode_case(), yolov8_product, yolov8_oracle),
        (retina_decode_case(), retina_product, retina_oracle),
        (rpn_decode_case(), rpn_decode_product, rpn_decode_oracle),
        (rpn_decode_case(batch=1, h=6, w=7, top_n=1000), rpn_decode_product, rpn_decode_oracle),  # fewer anchors than top_n
        (rpn_nms_case(), rpn_nms_product, rpn_nms_oracle),
        (roi_align_case(), roi_align_product, roi_align_oracle),
        (roi_align_case(P=16, C=4, res=7, ratio=2, seed=2), roi_align_product, roi_align_oracle),
        (predictor_decode_case(), predictor_decode_product, predictor_decode_oracle),
        (batched_nms_case(0), batched_nms_product, batched_nms_oracle),
        (batched_nms_case(1), batched_nms_product, batched_nms_oracle),
        (batched_nms_case(2), batched_nms_product, batched_nms_oracle),
        (batched_nms_case(0, class_mod=2, seed=16), batched_nms_product, batched_nms_oracle),
        (batched_nms_case(1, class_mod=2, seed=17), batched_nms_product, batched_nms_oracle),
        (batched_nms_case(2, class_mod=2, seed=18), batched_nms_product, batched_nms_oracle),
        (mask_case(), mask_product, mask_oracle),
        (yolov5_decode_case(), yolov5_product, yolov5_oracle),
        (yolov5_decode_case(batch=1, seed=14, size=320, seg=True), yolov5_product, yolov5_oracle),
        (mish_case(), mish_product, mish_oracle),
    ]


def run_reference(case, dev):
    """The reference's own plugin on the GPU, through the C-ABI v-table."""
    from oracle import ref
    creators = ref.load_plugins(case.family)

    def once(batch, inputs, out_shapes):
        v = ref.make_plugin(creators[case.plugin], blob=case.blob, fields=case.fields if case.blob is None else None)
        outs = ref.run_plugin(v, batch, [_t(x, dev) for x in inputs], out_shapes)
        return [o.cpu().numpy() for o in outs]

    if not case.per_image:
        return once(case.batch, case.inputs, case.out_shapes)
    per = [once(1, [x[b:b + 1] for x in case.inputs], [(1,) + tuple(s[1:]) for s in case.out_shapes]) for b in range(case.batch)]
    return [np.concatenate([p[k] for p in per], 0) for k in range(len(case.out_shapes))]


def compare(case, got, want, what):
    g, w = case.canon(got), case.canon(want)
    assert len(g) == len(w), what
    for k, (a, b) in enumerate(zip(g, w)):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape, f"{case.name} [{what}] output {k}: shape {a.shape} vs {b.shape}"
        if case.rtol == 0.0 and case.atol == 0.0:
            assert np.array_equal(a, b), f"{case.name} [{what}] output {k}: not bit-exact ({(a != b).sum()} of {a.size} differ)"
        else:
            assert np.allclose(a, b, rtol=case.rtol, atol=case.atol), f"{case.name} [{what}] output {k}: max abs diff {np.abs(a - b).max()}"
