"""ORACLE (test infrastructure): PyTorch-CPU fp32 restatements of the reference network builders,
layer by layer, loading the same seeded synthetic `.wts` the MI355X engine loads.

  LeNet-5     lenet/gen_wts.py:10-45  <->  lenet/lenet.cpp:36-155
  ResNet-50   resnet/resnet50.cpp:77-229  (BN eps 1e-5, stride on the 3x3, relu(conv3 + shortcut))
  YOLOv8n-det yolov8/src/block.cpp:45-257 + yolov8/src/model.cpp:9-25,98-310 (BN eps 1e-3, SiLU, C2F, SPPF,
              DFL = softmax(16) . arange via 1x1 conv); output = the three [B, 4+nc, g] plugin inputs.

Each model is a function over a `Params` provider: in "init" mode the provider creates seeded weights of
the shapes the forward pass asks for (that is how the synthetic .wts is generated); in "load" mode it
returns the tensors read back from a .wts file.  Real trained weights cannot be obtained offline, so the
weights are He-scaled random with near-identity BatchNorm statistics.

Parity status: parity unpinned (the reference ships no weights, goldens or tests).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


class Params:
    """name -> tensor provider.  init mode: creates + records; load mode: looks up (and checks shapes)."""

    def __init__(self, tensors=None, seed=0):
        self.load = tensors is not None
        self.t = OrderedDict()
        self.src = tensors
        self.g = torch.Generator().manual_seed(seed)

    def _get(self, name, shape, init):
        if self.load:
            v = torch.as_tensor(np.asarray(self.src[name], dtype=np.float32)).reshape(shape)
        else:
            v = init().float()
        self.t[name] = v
        return v

    def randn(self, *shape):
        return torch.randn(*shape, generator=self.g)

    def rand(self, *shape):
        return torch.rand(*shape, generator=self.g)

    def conv_w(self, name, cout, cin, k, gain=2.0):
        fan = cin * k * k
        return self._get(name, (cout, cin, k, k), lambda: self.randn(cout, cin, k, k) * math.sqrt(gain / fan))

    def vec(self, name, n, init):
        return self._get(name, (n,), init)

    def bn(self, prefix, c, gamma_scale=1.0):
        w = self.vec(prefix + ".weight", c, lambda: gamma_scale * (0.9 + 0.2 * self.rand(c)))
        b = self.vec(prefix + ".bias", c, lambda: 0.1 * self.randn(c))
        m = self.vec(prefix + ".running_mean", c, lambda: 0.1 * self.randn(c))
        v = self.vec(prefix + ".running_var", c, lambda: 0.8 + 0.4 * self.rand(c))
        # exporters also dump num_batches_tracked (a 0-dim tensor -> 1 value); keep it for format fidelity
        self._get(prefix + ".num_batches_tracked", (1,), lambda: torch.zeros(1))
        return w, b, m, v


# --------------------------------------------------------------------------------------------- LeNet-5

class Params64(Params):
    """load mode only: the same fp32 weights widened to float64 - the forward functions then evaluate the reference graph in double, the value an fp32
    implementation (this oracle's own fp32 pass included: 5e-5 on YOLOv8n's O(30) head logits at 640 x 640) is a rounding of."""

    def _get(self, name, shape, init):
        v = torch.as_tensor(np.asarray(self.src[name], dtype=np.float32)).double().reshape(shape)
        self.t[name] = v
        return v

def lenet(p: Params, x):
    """lenet/gen_wts.py:26-45 (max pooling; softmax appended as in lenet.cpp:133-137)."""
    y = F.conv2d(x, p._get("conv1.weight", (6, 1, 5, 5), lambda: p.randn(6, 1, 5, 5) * 0.2),
                 p.vec("conv1.bias", 6, lambda: 0.1 * p.randn(6)))
    y = F.max_pool2d(F.relu(y), 2, 2)
    y = F.conv2d(y, p._get("conv2.weight", (16, 6, 5, 5), lambda: p.randn(16, 6, 5, 5) * 0.08),
                 p.vec("conv2.bias", 16, lambda: 0.1 * p.randn(16)))
    y = F.max_pool2d(F.relu(y), 2, 2)
    y = y.reshape(y.shape[0], -1)
    for name, o, i, relu in (("fc1", 120, 400, True), ("fc2", 84, 120, True), ("fc3", 10, 84, False)):
        w = p._get(name + ".weight", (o, i), lambda o=o, i=i: p.randn(o, i) * math.sqrt(2.0 / i))
        b = p.vec(name + ".bias", o, lambda o=o: 0.1 * p.randn(o))
        y = F.linear(y, w, b)
        if relu:
            y = F.relu(y)
    return F.softmax(y, dim=1)


# ------------------------------------------------------------------------------------------- ResNet-50
def _bn_eval(x, stats, eps):
    w, b, m, v = stats
    return F.batch_norm(x, m, v, w, b, training=False, eps=eps)


def resnet50(p: Params, x):
    """resnet/resnet50.cpp:155-229; returns the FC-1000 logits [B, 1000] (no softmax, as the reference)."""
    eps = 1e-5

    def conv_bn(x, cname, bname, cout, k, s, pad):
        cin = x.shape[1]
        y = F.conv2d(x, p.conv_w(cname + ".weight", cout, cin, k), None, stride=s, padding=pad)
        return _bn_eval(y, p.bn(bname, cout), eps)

    def bottleneck(x, inch, outch, stride, l):
        a = F.relu(conv_bn(x, l + "conv1", l + "bn1", outch, 1, 1, 0))
        b = F.relu(conv_bn(a, l + "conv2", l + "bn2", outch, 3, stride, 1))
        c = conv_bn(b, l + "conv3", l + "bn3", outch * 4, 1, 1, 0)
        sc = x
        if stride != 1 or inch != outch * 4:
            sc = conv_bn(x, l + "downsample.0", l + "downsample.1", outch * 4, 1, stride, 0)
        return F.relu(sc + c)

    y = F.relu(conv_bn(x, "conv1", "bn1", 64, 7, 2, 3))
    y = F.max_pool2d(y, 3, 2, 1)
    inch = 64
    for stage, nblk in enumerate((3, 4, 6, 3)):
        width = 64 << stage
        for b in range(nblk):
            stride = 2 if (b == 0 and stage > 0) else 1
            y = bottleneck(y, inch, width, stride, f"layer{stage + 1}.{b}.")
            inch = width * 4
    y = F.avg_pool2d(y, y.shape[-1], 1).flatten(1)
    w = p._get("fc.weight", (1000, 2048), lambda: p.randn(1000, 2048) * math.sqrt(1.0 / 2048))
    b = p.vec("fc.bias", 1000, lambda: 0.1 * p.randn(1000))
    return F.linear(y, w, b)


# ------------------------------------------------------------------------------------------ YOLOv8-det
def _get_width(x, gw, max_channels, divisor=8):  # model.cpp:13-16
    ch = int(math.ceil((x * gw) / divisor)) * divisor
    return max_channels if ch >= max_channels else ch


def _get_depth(x, gd):  # model.cpp:18-25
    if x == 1:
        return 1
    r = int(math.floor(x * gd + 0.5))  # C round(): half away from zero
    if x * gd - int(x * gd) == 0.5 and int(x * gd) % 2 == 0:
        r -= 1
    return max(r, 1)


def yolov8_det(p: Params, x, num_class=80, gd=0.33, gw=0.25, max_channels=1024, cls_bias=-7.0, cls_gain=80.0, task="det", nk=17):
    """Returns ([B, 4+nc(+extra), g] for stride 8/16/32, strides) — exactly what feeds YoloLayer_TRT.  task "seg" / "pose" / "obb"
    adds the cv4 branch of buildEngineYolov8Seg / Pose / Obb (model.cpp:54-96, 1253-1272, 1483-1535, 2699-2721): 32 mask
    coefficients, nk*3 keypoint values or 1 angle logit per cell, appended after the class rows; "seg" also returns the Proto
    tensor [B, 32, H/4, W/4] (model.cpp:36-52) as a third value."""
    eps = 1e-3
    extra = {"det": 0, "seg": 32, "pose": nk * 3, "obb": 1}[task]

    def cbs(x, name, cout, k, s, pad):  # convBnSiLU, block.cpp:79-96
        y = F.conv2d(x, p.conv_w(name + ".conv.weight", cout, x.shape[1], k), None, stride=s, padding=pad)
        y = _bn_eval(y, p.bn(name + ".bn", cout), eps)
        return y * torch.sigmoid(y)

    def bottleneck(x, c1, c2, shortcut, name):  # block.cpp:98-110
        y = cbs(cbs(x, name + ".cv1", c2, 3, 1, 1), name + ".cv2", c2, 3, 1, 1)
        return x + y if (shortcut and c1 == c2) else y

    def c2f(x, c2, n, shortcut, name, e=0.5):  # block.cpp:126-155
        c_ = int(float(c2) * e)
        y = cbs(x, name + ".cv1", 2 * c_, 1, 1, 0)
        parts = [y[:, :c_], y[:, c_:]]
        cur = parts[1]
        for i in range(n):
            cur = bottleneck(cur, c_, c_, shortcut, f"{name}.m.{i}")
            parts.append(cur)
        return cbs(torch.cat(parts, 1), name + ".cv2", c2, 1, 1, 0)

    def sppf(x, c1, c2, k, name):  # block.cpp:214-237
        y = cbs(x, name + ".cv1", c1 // 2, 1, 1, 0)
        ps = [y]
        for _ in range(3):
            ps.append(F.max_pool2d(ps[-1], k, 1, k // 2))
        return cbs(torch.cat(ps, 1), name + ".cv2", c2, 1, 1, 0)

    def up(x):
        return F.interpolate(x, scale_factor=2, mode="nearest")

    W = lambda v: _get_width(v, gw, max_channels)  # noqa: E731
    H_in, W_in = x.shape[2], x.shape[3]
    p1 = cbs(x, "model.0", W(64), 3, 2, 1)
    p2 = cbs(p1, "model.1", W(128), 3, 2, 1)
    c2 = c2f(p2, W(128), _get_depth(3, gd), True, "model.2")
    p3 = cbs(c2, "model.3", W(256), 3, 2, 1)
    c4 = c2f(p3, W(256), _get_depth(6, gd), True, "model.4")
    p4 = cbs(c4, "model.5", W(512), 3, 2, 1)
    c6 = c2f(p4, W(512), _get_depth(6, gd), True, "model.6")
    p5 = cbs(c6, "model.7", W(1024), 3, 2, 1)
    c8 = c2f(p5, W(1024), _get_depth(3, gd), True, "model.8")
    c9 = sppf(c8, W(1024), W(1024), 5, "model.9")
    c12 = c2f(torch.cat([up(c9), c6], 1), W(512), _get_depth(3, gd), False, "model.12")
    c15 = c2f(torch.cat([up(c12), c4], 1), W(256), _get_depth(3, gd), False, "model.15")
    c16 = cbs(c15, "model.16", W(256), 3, 2, 1)
    c18 = c2f(torch.cat([c16, c12], 1), W(512), _get_depth(3, gd), False, "model.18")
    c19 = cbs(c18, "model.19", W(512), 3, 2, 1)
    c21 = c2f(torch.cat([c19, c9], 1), W(1024), _get_depth(3, gd), False, "model.21")

    base_in = 80 if gw == 1.25 else 64
    base_out = max(64, min(num_class, 100)) if gw == 0.25 else W(256)
    strides = [H_in // t.shape[2] for t in (p3, p4, p5)]
    outs = []
    dfl_w = None
    for lv, feat in enumerate((c15, c18, c21)):
        s = str(lv)
        b = cbs(cbs(feat, f"model.22.cv2.{s}.0", base_in, 3, 1, 1), f"model.22.cv2.{s}.1", base_in, 3, 1, 1)
        box = F.conv2d(b, p.conv_w(f"model.22.cv2.{s}.2.weight", 64, base_in, 1, gain=4.0),
                       p.vec(f"model.22.cv2.{s}.2.bias", 64, lambda: 1.0 + 0.1 * p.randn(64)))
        k = cbs(cbs(feat, f"model.22.cv3.{s}.0", base_out, 3, 1, 1), f"model.22.cv3.{s}.1", base_out, 3, 1, 1)
        cls = F.conv2d(k, p.conv_w(f"model.22.cv3.{s}.2.weight", num_class, base_out, 1, gain=cls_gain),
                       p.vec(f"model.22.cv3.{s}.2.bias", num_class, lambda: cls_bias + 0.1 * p.randn(num_class)))
        cat = torch.cat([box, cls], 1)
        g = cat.shape[2] * cat.shape[3]
        flat = cat.reshape(cat.shape[0], 64 + num_class, g)
        boxp, clsp = flat[:, :64], flat[:, 64:]
        if dfl_w is None:
            dfl_w = p._get("model.22.dfl.conv.weight", (1, 16, 1, 1), lambda: torch.arange(16.0).reshape(1, 16, 1, 1))
        # DFL, block.cpp:239-257: (64,g)->(4,16,g)->(16,4,g) softmax over the 16 bins, 1x1 conv(arange)
        t = boxp.reshape(-1, 4, 16, g).permute(0, 2, 1, 3)
        t = F.conv2d(F.softmax(t, dim=1), dfl_w).reshape(-1, 4, g)
        parts = [t, clsp]
        if extra:  # cv4_conv_combined: two 3x3 convBnSiLU + biased 1x1 conv, flattened to (extra, g)
            mid = max(c15.shape[1] // 4, extra)  # ultralytics Segment/Pose/OBB: c4 = max(ch[0] // 4, n) (== the table of model.cpp:61-70)
            e = cbs(cbs(feat, f"model.22.cv4.{s}.0", mid, 3, 1, 1), f"model.22.cv4.{s}.1", mid, 3, 1, 1)
            e = F.conv2d(e, p.conv_w(f"model.22.cv4.{s}.2.weight", extra, mid, 1), p.vec(f"model.22.cv4.{s}.2.bias", extra, lambda: 0.1 * p.randn(extra)))
            parts.append(e.reshape(e.shape[0], extra, g))
        outs.append(torch.cat(parts, 1).contiguous())
    if task == "seg":  # Proto: cv1 3x3 -> ConvTranspose2d(2, 2, bias) -> cv2 3x3 -> cv3 1x1 to 32 masks
        mid = W(256)
        y = cbs(c15, "model.22.proto.cv1", mid, 3, 1, 1)
        wt = p._get("model.22.proto.upsample.weight", (mid, mid, 2, 2), lambda: p.randn(mid, mid, 2, 2) * math.sqrt(2.0 / mid))
        y = F.conv_transpose2d(y, wt, p.vec("model.22.proto.upsample.bias", mid, lambda: 0.1 * p.randn(mid)), stride=2)
        y = cbs(cbs(y, "model.22.proto.cv2", mid, 3, 1, 1), "model.22.proto.cv3", 32, 1, 1, 0)
        return outs, strides, y
    return outs, strides


# ---------------------------------------------------------------------------------------- RetinaFace-R50
def retinaface_r50(p: Params, x, head_gain=0.5):
    """retinaface/retina_r50.cpp:100-212.  Returns the three [B, 32, h, w] plugin inputs
    (bbox 8 | class 4 | landmark 20) for stride 8/16/32."""
    eps = 1e-5

    def conv_bn(x, cname, bname, cout, k, s, pad):
        y = F.conv2d(x, p.conv_w(cname + ".weight", cout, x.shape[1], k), None, stride=s, padding=pad)
        # synthetic weights: a small gamma on the last BN of every residual branch keeps activations O(1-10)
        # through 16 bottlenecks so that fp16 storage (and the exp() in the decode) stay in range
        return _bn_eval(y, p.bn(bname, cout, 0.25 if bname.endswith("bn3") else 1.0), eps)

    def bottleneck(x, inch, outch, stride, l):
        a = F.relu(conv_bn(x, l + "conv1", l + "bn1", outch, 1, 1, 0))
        b = F.relu(conv_bn(a, l + "conv2", l + "bn2", outch, 3, stride, 1))
        c = conv_bn(b, l + "conv3", l + "bn3", outch * 4, 1, 1, 0)
        sc = x
        if stride != 1 or inch != outch * 4:
            sc = conv_bn(x, l + "downsample.0", l + "downsample.1", outch * 4, 1, stride, 0)
        return F.relu(sc + c)

    def cbr(x, cout, k, pad, relu, name):  # conv_bn_relu :70-85
        y = conv_bn(x, name + ".0", name + ".1", cout, k, 1, pad)
        return F.relu(y) if relu else y

    def ssh(x, l):  # :87-98
        c3 = cbr(x, 128, 3, 1, False, l + ".conv3X3")
        c5a = cbr(x, 64, 3, 1, True, l + ".conv5X5_1")
        c5 = cbr(c5a, 64, 3, 1, False, l + ".conv5X5_2")
        c7 = cbr(cbr(c5a, 64, 3, 1, True, l + ".conv7X7_2"), 64, 3, 1, False, l + ".conv7x7_3")
        return F.relu(torch.cat([c3, c5, c7], 1))

    y = F.relu(conv_bn(x, "body.conv1", "body.bn1", 64, 7, 2, 3))
    y = F.max_pool2d(y, 3, 2, 1)
    inch = 64
    stages = []
    for stage, nblk in enumerate((3, 4, 6, 3)):
        width = 64 << stage
        for b in range(nblk):
            y = bottleneck(y, inch, width, 2 if (b == 0 and stage > 0) else 1, f"body.layer{stage + 1}.{b}.")
            inch = width * 4
        stages.append(y)
    o1 = cbr(stages[1], 256, 1, 0, True, "fpn.output1")
    o2 = cbr(stages[2], 256, 1, 0, True, "fpn.output2")
    o3 = cbr(stages[3], 256, 1, 0, True, "fpn.output3")
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")  # grouped 2x2/2 deconv of ones, :157-166  # noqa: E731
    o2 = cbr(o2 + up(o3), 256, 3, 1, True, "fpn.merge2")
    o1 = cbr(o1 + up(o2), 256, 3, 1, True, "fpn.merge1")
    outs = []
    for l, f in enumerate((ssh(o1, "ssh1"), ssh(o2, "ssh2"), ssh(o3, "ssh3"))):
        parts = []
        for name, ch in (("BboxHead", 8), ("ClassHead", 4), ("LandmarkHead", 20)):
            w = p.conv_w(f"{name}.{l}.conv1x1.weight", ch, 256, 1, gain=head_gain)
            b = p.vec(f"{name}.{l}.conv1x1.bias", ch, lambda ch=ch: 0.1 * p.randn(ch))
            parts.append(F.conv2d(f, w, b))
        outs.append(torch.cat(parts, 1).contiguous())
    return outs


def make_weights(model_fn, example_input, seed=0, **kw):
    """Run `model_fn` in init mode on `example_input`; returns (OrderedDict name -> tensor, output)."""
    p = Params(None, seed)
    with torch.inference_mode():
        out = model_fn(p, example_input, **kw)
    return p.t, out


# ---------------------------------------------------------------------------------------------------------
RCNN_DEFAULTS = dict(num_classes=80, anchor_sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0), pre_nms_topk=6000,
                     rpn_nms_thresh=0.7, post_nms_topk=1000, stride=16, sampling_ratio=0, pooler_resolution=14,
                     nms_thresh_test=0.5, detections_per_image=100, bbox_reg_weights=(10.0, 10.0, 5.0, 5.0), nms_method=1,
                     pixel_mean=(103.53, 116.28, 123.675), pixel_std=(1.0, 1.0, 1.0))


def rcnn_r50c4(p: Params, x_hwc, stage="all", given=None, **cfg):
    """Faster R-CNN R50-C4, rcnn/rcnn.cpp:79-278 + rcnn/backbone.hpp:26-229 (detectron2 export after fuse-bn: every
    conv has a bias, stride in the first 1x1).  x_hwc: [B, H, W, 3] fp32.  Returns a dict with every stage:
    features, logits, deltas, rpn_scores, rpn_boxes, proposals, roi (RoiAlign out), cls_prob, box_deltas,
    pred_{scores,boxes,classes}, scores, boxes, labels.  The plugin stages use the C restatements (det_post).

    `given` lets a parity test restart the chain from tensors produced elsewhere (e.g. the GPU engine's own
    features/proposals), so that discrete top-k / NMS decisions upstream do not mask a downstream comparison.
    stage="init" touches every weight with a cheap forward (used to generate the synthetic .wts)."""
    from . import det_post
    c = dict(RCNN_DEFAULTS)
    c.update(cfg)
    given = given or {}
    out = {}

    def conv(x, name, cout, k, s, pad, relu, gain=2.0):
        w = p.conv_w(name + ".weight", cout, x.shape[1], k, gain=gain)
        b = p.vec(name + ".bias", cout, lambda: 0.05 * p.randn(cout))
        y = F.conv2d(x, w, b, stride=s, padding=pad)
        return F.relu(y) if relu else y

    def block(x, inch, mid, outch, stride, l):  # backbone.hpp:104-169
        a = conv(x, l + ".conv1", mid, 1, stride, 0, True)
        b = conv(a, l + ".conv2", mid, 3, 1, 1, True)
        d = conv(b, l + ".conv3", outch, 1, 1, 0, False, gain=0.125)  # synthetic: small residual branch, fp16-safe depth
        sc = conv(x, l + ".shortcut", outch, 1, stride, 0, False, gain=1.0) if inch != outch else x
        return F.relu(d + sc)

    def stage_(x, n, inch, mid, outch, first_stride, l):
        for i in range(n):
            x = block(x, inch, mid, outch, first_stride if i == 0 else 1, f"{l}.{i}")
            inch = outch
        return x

    B, H, W, _ = x_hwc.shape
    A = len(c["anchor_sizes"]) * len(c["aspect_ratios"])
    nc = c["num_classes"]
    if "features" in given:
        feats = torch.as_tensor(given["features"])
    else:
        x = x_hwc.permute(0, 3, 1, 2)
        x = (x - torch.tensor(c["pixel_mean"]).view(1, 3, 1, 1)) / torch.tensor(c["pixel_std"]).view(1, 3, 1, 1)
        # synthetic stem gain: the (x - mean) input is O(100); bring the stem output to O(1)
        y = conv(x, "backbone.stem.conv1", 64, 7, 2, 3, True, gain=2.0 / 70.0 ** 2)
        y = F.max_pool2d(y, 3, 2, 1)
        inch, mid, outch = 64, 64, 256
        for s, n in enumerate((3, 4, 6)):
            y = stage_(y, n, inch, mid, outch, 1 if s == 0 else 2, f"backbone.res{s + 2}")
            inch, mid, outch = outch, mid * 2, outch * 2
        feats = y
    out["features"] = feats
    if stage == "backbone":
        return out
    fh, fw = feats.shape[-2:]
    hid = conv(feats, "proposal_generator.rpn_head.conv", feats.shape[1], 3, 1, 1, True)
    out["logits"] = conv(hid, "proposal_generator.rpn_head.objectness_logits", A, 1, 1, 0, False, gain=8.0)
    out["deltas"] = conv(hid, "proposal_generator.rpn_head.anchor_deltas", 4 * A, 1, 1, 0, False, gain=0.05)
    if stage == "init":
        pre = post = 8
    else:
        pre, post = c["pre_nms_topk"], c["post_nms_topk"]
    if "proposals" in given:
        proposals = np.asarray(given["proposals"], np.float32)
    else:
        anchors = det_post.generate_anchors(c["anchor_sizes"], c["aspect_ratios"])
        rs, rb = det_post.rpn_decode(out["logits"].numpy(), out["deltas"].numpy(), fh, fw, H, W, float(c["stride"]), anchors, pre)
        out["rpn_scores"], out["rpn_boxes"] = rs, rb
        proposals = det_post.rpn_nms(rs, rb, post, c["rpn_nms_thresh"])
    out["proposals"] = proposals
    P = proposals.shape[1]
    roi = det_post.roi_align(proposals, feats.numpy(), c["pooler_resolution"], 1.0 / c["stride"], c["sampling_ratio"])
    out["roi"] = roi
    r = torch.from_numpy(roi).reshape(B * P, feats.shape[1], c["pooler_resolution"], c["pooler_resolution"])
    r = stage_(r, 3, feats.shape[1], 512, 2048, 2, "roi_heads.res5")
    pooled = r.mean((2, 3))
    wc = p._get("roi_heads.box_predictor.cls_score.weight", (nc + 1, 2048), lambda: p.randn(nc + 1, 2048) * 0.06)
    bc = p.vec("roi_heads.box_predictor.cls_score.bias", nc + 1, lambda: 0.1 * p.randn(nc + 1))
    wb = p._get("roi_heads.box_predictor.bbox_pred.weight", (4 * nc, 2048), lambda: p.randn(4 * nc, 2048) * 0.01)
    bb = p.vec("roi_heads.box_predictor.bbox_pred.bias", 4 * nc, lambda: 0.01 * p.randn(4 * nc))
    prob = F.softmax(F.linear(pooled, wc, bc), dim=1)
    out["cls_prob"] = prob.reshape(B, P, nc + 1)
    out["box_deltas"] = F.linear(pooled, wb, bb).reshape(B, P, 4 * nc)
    fg = out["cls_prob"][:, :, :nc].contiguous().numpy()
    ps, pb, pc = det_post.predictor_decode(fg, out["box_deltas"].numpy(), proposals, H, W, c["bbox_reg_weights"])
    out["pred_scores"], out["pred_boxes"], out["pred_classes"] = ps, pb, pc
    dets = min(c["detections_per_image"], P)
    out["scores"], out["boxes"], out["labels"] = det_post.batched_nms(c["nms_method"], ps, pb, pc, dets, c["nms_thresh_test"])
    if c.get("mask_on") or stage == "init":
        # MaskHead, rcnn.cpp:202-232: RoIAlign of the final boxes -> res5 (same weights) -> ConvTranspose 2x2/2 + ReLU ->
        # 1x1 predictor -> the plane of each detection's own class through a sigmoid (MaskRcnnInference.cu:8-30)
        boxes = np.asarray(given.get("boxes", out["boxes"]), np.float32)
        labels = np.asarray(given.get("labels", out["labels"]), np.float32)
        D = boxes.shape[1]
        mroi = det_post.roi_align(boxes, feats.numpy(), c["pooler_resolution"], 1.0 / c["stride"], c["sampling_ratio"])
        m = torch.from_numpy(mroi).reshape(B * D, feats.shape[1], c["pooler_resolution"], c["pooler_resolution"])
        m = stage_(m, 3, feats.shape[1], 512, 2048, 2, "roi_heads.res5")
        wd = p._get("roi_heads.mask_head.deconv.weight", (2048, 256, 2, 2), lambda: p.randn(2048, 256, 2, 2) * math.sqrt(2.0 / 2048))
        bd = p.vec("roi_heads.mask_head.deconv.bias", 256, lambda: 0.05 * p.randn(256))
        m = F.relu(F.conv_transpose2d(m, wd, bd, stride=2))
        logits = conv(m, "roi_heads.mask_head.predictor", nc, 1, 1, 0, False, gain=4.0)
        out["mask_logits"] = logits.reshape(B, D, nc, *logits.shape[-2:])
        out["masks"] = det_post.mask_select(labels, out["mask_logits"].numpy())
    return out
