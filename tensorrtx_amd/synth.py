"""Seeded synthetic inputs of the shapes SURVEY.md §8(d) prescribes (no datasets / weights offline)."""
import numpy as np


def yolo_head_tensors(batch, classes=80, net_h=640, net_w=640, strides=(8, 16, 32), objects=(40, 180), seed=0,
                      bg_mean=-8.0, bg_std=1.5):
    """Planted-object head tensors for the YoloLayer plugin: list of [B, 4+classes, gh*gw] fp32.

    Box channels ltrb ~ U(0, 10) cells, class logits ~ N(bg_mean, bg_std^2); per image K in `objects`
    planted objects: one class gets logit +4..+8 on 3-8 neighbouring cells with consistent boxes so
    that NMS clusters exist (candidates(sigma >= 0.1) << 1000, survivors(conf > 0.5) ~ 100-600).
    """
    rng = np.random.default_rng(seed)
    outs = []
    for s in strides:
        gh, gw = net_h // s, net_w // s
        x = np.empty((batch, 4 + classes, gh * gw), dtype=np.float32)
        x[:, :4] = rng.uniform(0.0, 10.0, size=(batch, 4, gh * gw)).astype(np.float32)
        x[:, 4:] = rng.normal(bg_mean, bg_std, size=(batch, classes, gh * gw)).astype(np.float32)
        outs.append(x)
    for b in range(batch):
        k_obj = int(rng.integers(objects[0], objects[1] + 1))
        for _ in range(k_obj):
            l = int(rng.integers(0, len(strides)))
            s = strides[l]
            gh, gw = net_h // s, net_w // s
            row, col = int(rng.integers(1, gh - 1)), int(rng.integers(1, gw - 1))
            cls = int(rng.integers(0, classes))
            # object box in pixels around the anchor point
            cx, cy = (col + 0.5) * s, (row + 0.5) * s
            hw_, hh_ = rng.uniform(1.0, 6.0) * s, rng.uniform(1.0, 6.0) * s
            x1, y1, x2, y2 = cx - hw_, cy - hh_, cx + hw_, cy + hh_
            n_cells = int(rng.integers(3, 9))
            for _c in range(n_cells):
                r = min(max(row + int(rng.integers(-1, 2)), 0), gh - 1)
                c = min(max(col + int(rng.integers(-1, 2)), 0), gw - 1)
                e = r * gw + c
                jit = rng.normal(0.0, 0.15, size=4)
                outs[l][b, 0, e] = (c + 0.5) - x1 / s + jit[0]
                outs[l][b, 1, e] = (r + 0.5) - y1 / s + jit[1]
                outs[l][b, 2, e] = x2 / s - (c + 0.5) + jit[2]
                outs[l][b, 3, e] = y2 / s - (r + 0.5) + jit[3]
                outs[l][b, 4 + cls, e] = rng.uniform(0.5, 8.0)
    return outs


def images(batch, h=640, w=640, seed=0, n_rect=(8, 30)):
    """Synthetic RGB images in [0, 1], [B, 3, H, W] fp32: dark noisy background with random bright
    rectangles, so that a randomly initialised detector sees spatially varying features
    (uniform noise alone averages out to a constant feature map)."""
    rng = np.random.default_rng(seed)
    img = rng.uniform(0.0, 0.15, size=(batch, 3, h, w)).astype(np.float32)
    for b in range(batch):
        for _ in range(int(rng.integers(n_rect[0], n_rect[1] + 1))):
            rw, rh = int(rng.integers(w // 40, w // 4)), int(rng.integers(h // 40, h // 4))
            x0, y0 = int(rng.integers(0, w - rw)), int(rng.integers(0, h - rh))
            col = rng.uniform(0.2, 1.0, size=(3, 1, 1)).astype(np.float32)
            img[b, :, y0:y0 + rh, x0:x0 + rw] = col + rng.normal(0, 0.03, size=(3, rh, rw)).astype(np.float32)
    return np.clip(img, 0.0, 1.0)


def retina_head_tensors(batch, net_h, net_w, faces=200, seed=0):
    """RetinaFace plugin inputs: list of [B, 32, h*w] fp32 for stride 8/16/32 =
    bbox(2x4) | cls(2x2) | landmark(2x10) planes; cls pair difference ~ N(-5, 2^2), deltas ~ N(0, 1),
    ~`faces` planted high-confidence anchors per image with clustered neighbours."""
    rng = np.random.default_rng(seed)
    outs = []
    for s in (8, 16, 32):
        h, w = net_h // s, net_w // s
        x = rng.normal(0.0, 1.0, size=(batch, 32, h * w)).astype(np.float32)
        diff = rng.normal(-5.0, 2.0, size=(batch, 2, h * w)).astype(np.float32)
        base = rng.normal(0.0, 1.0, size=(batch, 2, h * w)).astype(np.float32)
        x[:, 8] = base[:, 0]; x[:, 9] = base[:, 0] + diff[:, 0]      # k = 0: (conf1, conf2)
        x[:, 10] = base[:, 1]; x[:, 11] = base[:, 1] + diff[:, 1]    # k = 1
        outs.append(x)
    for b in range(batch):
        for _ in range(faces):
            l = int(rng.integers(0, 3))
            h, w = net_h // (8 << l), net_w // (8 << l)
            cy, cx, k = int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(0, 2))
            for dy, dx in ((0, 0), (0, 1), (1, 0)):
                y, x_ = min(cy + dy, h - 1), min(cx + dx, w - 1)
                e = y * w + x_
                outs[l][b, 8 + 2 * k + 1, e] = outs[l][b, 8 + 2 * k, e] + rng.uniform(1.0, 6.0)
                outs[l][b, 4 * k:4 * k + 4, e] = rng.normal(0, 0.3, size=4)
    return outs


def rcnn_rpn_tensors(batch, anchors=15, h=50, w=84, seed=0):
    """RPN head outputs (C5): logits ~ N(0, 2^2) [B, A, h, w], deltas ~ N(0, 0.5^2) [B, 4A, h, w]."""
    rng = np.random.default_rng(seed)
    return (rng.normal(0, 2.0, size=(batch, anchors, h, w)).astype(np.float32),
            rng.normal(0, 0.5, size=(batch, anchors * 4, h, w)).astype(np.float32))


def rcnn_box_head_tensors(batch, n=1000, classes=80, img_h=800, img_w=1333, seed=0):
    """Box-head outputs (C5): softmax scores from logits ~ N(0, 3^2) over classes+1 (background dropped),
    deltas ~ N(0, 1) [B, N, C, 4], proposals = random boxes inside the image."""
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 3.0, size=(batch, n, classes + 1))
    e = np.exp(logits - logits.max(-1, keepdims=True))
    scores = (e / e.sum(-1, keepdims=True))[..., :classes].astype(np.float32)
    deltas = rng.normal(0, 1.0, size=(batch, n, classes, 4)).astype(np.float32)
    x1 = rng.uniform(0, img_w - 40, size=(batch, n)); y1 = rng.uniform(0, img_h - 40, size=(batch, n))
    bw = rng.uniform(8, 300, size=(batch, n)); bh = rng.uniform(8, 300, size=(batch, n))
    props = np.stack([x1, y1, np.minimum(x1 + bw, img_w), np.minimum(y1 + bh, img_h)], -1).astype(np.float32)
    return scores, deltas, props


def yolov8n_state(seed=0, num_class=80):
    """Seeded synthetic weights of YOLOv8n-det under the reference's `.wts` key names (ultralytics state_dict keys, as
    read by yolov8/src/block.cpp:79-257 / model.cpp:98-310): OrderedDict name -> fp32 array.  Trained weights cannot be
    obtained offline; the values are He-scaled with near-identity BatchNorm statistics, the class head has gain 80 and
    bias -7 so that a few hundred cells per image pass the 0.1 confidence gate.  Used by bench.py (product side: no oracle
    involved).  Draw order and distributions are those of the test-suite's generator, so both produce the same file
    (tests/test_runtime_cpu.py::test_product_side_yolov8n_weights_match_the_test_generator)."""
    import math
    from collections import OrderedDict

    import torch
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    randn = lambda *shape: torch.randn(*shape, generator=g)  # noqa: E731
    rand = lambda *shape: torch.rand(*shape, generator=g)    # noqa: E731

    def conv(name, cout, cin, k, gain=2.0):
        sd[name + ".weight"] = (randn(cout, cin, k, k) * math.sqrt(gain / (cin * k * k))).float()

    def cbs(name, cout, cin, k):  # Conv + BatchNorm (+ SiLU)
        conv(name + ".conv", cout, cin, k)
        sd[name + ".bn.weight"] = (1.0 * (0.9 + 0.2 * rand(cout))).float()
        sd[name + ".bn.bias"] = (0.1 * randn(cout)).float()
        sd[name + ".bn.running_mean"] = (0.1 * randn(cout)).float()
        sd[name + ".bn.running_var"] = (0.8 + 0.4 * rand(cout)).float()
        sd[name + ".bn.num_batches_tracked"] = torch.zeros(1)

    def c2f(name, cin, c2, n):
        c_ = c2 // 2
        cbs(name + ".cv1", 2 * c_, cin, 1)
        for i in range(n):
            cbs(f"{name}.m.{i}.cv1", c_, c_, 3)
            cbs(f"{name}.m.{i}.cv2", c_, c_, 3)
        cbs(name + ".cv2", c2, (2 + n) * c_, 1)

    cbs("model.0", 16, 3, 3)
    cbs("model.1", 32, 16, 3)
    c2f("model.2", 32, 32, 1)
    cbs("model.3", 64, 32, 3)
    c2f("model.4", 64, 64, 2)
    cbs("model.5", 128, 64, 3)
    c2f("model.6", 128, 128, 2)
    cbs("model.7", 256, 128, 3)
    c2f("model.8", 256, 256, 1)
    cbs("model.9.cv1", 128, 256, 1)
    cbs("model.9.cv2", 256, 512, 1)
    c2f("model.12", 384, 128, 1)
    c2f("model.15", 192, 64, 1)
    cbs("model.16", 64, 64, 3)
    c2f("model.18", 192, 128, 1)
    cbs("model.19", 128, 128, 3)
    c2f("model.21", 384, 256, 1)
    c3 = max(64, min(num_class, 100))
    for lv, cin in enumerate((64, 128, 256)):
        cbs(f"model.22.cv2.{lv}.0", 64, cin, 3)
        cbs(f"model.22.cv2.{lv}.1", 64, 64, 3)
        conv(f"model.22.cv2.{lv}.2", 64, 64, 1, gain=4.0)
        sd[f"model.22.cv2.{lv}.2.bias"] = (1.0 + 0.1 * randn(64)).float()
        cbs(f"model.22.cv3.{lv}.0", c3, cin, 3)
        cbs(f"model.22.cv3.{lv}.1", c3, c3, 3)
        conv(f"model.22.cv3.{lv}.2", num_class, c3, 1, gain=80.0)
        sd[f"model.22.cv3.{lv}.2.bias"] = (-7.0 + 0.1 * randn(num_class)).float()
        if lv == 0:
            sd["model.22.dfl.conv.weight"] = torch.arange(16.0).reshape(1, 16, 1, 1)
    return OrderedDict((k, v.numpy()) for k, v in sd.items())


class _Draw:
    """Seeded tensor provider of the synthetic-weight generators below: the draw order and distributions are those of the test-suite's
    generator (oracle/models_torch.py run in init mode), so both write the same file - tests/test_runtime_cpu.py asserts it per model.
    A name drawn twice keeps its first position and its LAST value (the R-CNN generator touches res5 from the box and the mask branch)."""

    def __init__(self, seed):
        import torch
        from collections import OrderedDict
        self.torch = torch
        self.g = torch.Generator().manual_seed(seed)
        self.sd = OrderedDict()

    def randn(self, *shape):
        return self.torch.randn(*shape, generator=self.g)

    def rand(self, *shape):
        return self.torch.rand(*shape, generator=self.g)

    def conv_w(self, name, cout, cin, k, gain=2.0):
        import math
        self.sd[name] = (self.randn(cout, cin, k, k) * math.sqrt(gain / (cin * k * k))).float()

    def bn(self, prefix, c, gamma_scale=1.0):
        self.sd[prefix + ".weight"] = (gamma_scale * (0.9 + 0.2 * self.rand(c))).float()
        self.sd[prefix + ".bias"] = (0.1 * self.randn(c)).float()
        self.sd[prefix + ".running_mean"] = (0.1 * self.randn(c)).float()
        self.sd[prefix + ".running_var"] = (0.8 + 0.4 * self.rand(c)).float()
        self.sd[prefix + ".num_batches_tracked"] = self.torch.zeros(1)

    def state(self):
        from collections import OrderedDict
        return OrderedDict((k, v.numpy()) for k, v in self.sd.items())


def _resnet50_body(d, prefix, bn3_gamma):
    """conv1 / bn1 + the 16 bottlenecks of torchvision's ResNet-50 naming (resnet/resnet50.cpp:155-229, retinaface/retina_r50.cpp:100-140)"""
    def conv_bn(cname, bname, cout, cin, k):
        d.conv_w(cname + ".weight", cout, cin, k)
        d.bn(bname, cout, bn3_gamma if bname.endswith("bn3") else 1.0)
    conv_bn(prefix + "conv1", prefix + "bn1", 64, 3, 7)
    inch = 64
    for stage, nblk in enumerate((3, 4, 6, 3)):
        width = 64 << stage
        for b in range(nblk):
            l = f"{prefix}layer{stage + 1}.{b}."
            stride = 2 if (b == 0 and stage > 0) else 1
            conv_bn(l + "conv1", l + "bn1", width, inch, 1)
            conv_bn(l + "conv2", l + "bn2", width, width, 3)
            conv_bn(l + "conv3", l + "bn3", width * 4, width, 1)
            if stride != 1 or inch != width * 4:
                conv_bn(l + "downsample.0", l + "downsample.1", width * 4, inch, 1)
            inch = width * 4


def resnet50_state(seed=0):
    """Seeded synthetic ResNet-50 weights under the keys resnet/resnet50.cpp reads (torchvision state_dict): OrderedDict name -> fp32 array."""
    import math
    d = _Draw(seed)
    _resnet50_body(d, "", 1.0)
    d.sd["fc.weight"] = (d.randn(1000, 2048) * math.sqrt(1.0 / 2048)).float()
    d.sd["fc.bias"] = (0.1 * d.randn(1000)).float()
    return d.state()


def retinaface_r50_state(seed=0, head_gain=0.5):
    """Seeded synthetic RetinaFace-R50 weights (retinaface/retina_r50.cpp:100-212 key names).  The last BatchNorm of every residual branch has a
    small gamma so that activations stay O(1-10) through 16 bottlenecks (fp16 storage, the exp() of the decode)."""
    d = _Draw(seed)
    _resnet50_body(d, "body.", 0.25)

    def cbr(name, cout, cin, k):
        d.conv_w(name + ".0.weight", cout, cin, k)
        d.bn(name + ".1", cout)
    cbr("fpn.output1", 256, 512, 1)
    cbr("fpn.output2", 256, 1024, 1)
    cbr("fpn.output3", 256, 2048, 1)
    cbr("fpn.merge2", 256, 256, 3)
    cbr("fpn.merge1", 256, 256, 3)
    for l in ("ssh1", "ssh2", "ssh3"):
        cbr(l + ".conv3X3", 128, 256, 3)
        cbr(l + ".conv5X5_1", 64, 256, 3)
        cbr(l + ".conv5X5_2", 64, 64, 3)
        cbr(l + ".conv7X7_2", 64, 64, 3)
        cbr(l + ".conv7x7_3", 64, 64, 3)
    for l in range(3):
        for name, ch in (("BboxHead", 8), ("ClassHead", 4), ("LandmarkHead", 20)):
            d.conv_w(f"{name}.{l}.conv1x1.weight", ch, 256, 1, gain=head_gain)
            d.sd[f"{name}.{l}.conv1x1.bias"] = (0.1 * d.randn(ch)).float()
    return d.state()


def rcnn_r50c4_state(seed=0, num_classes=80, anchors=15):
    """Seeded synthetic Faster / Mask R-CNN R50-C4 weights (detectron2 export after fuse-bn, the keys of rcnn/rcnn.cpp:79-278 and
    rcnn/backbone.hpp:26-229): every conv has a bias; small gains on the residual branches and the stem bring the (x - mean) input to O(1)."""
    import math
    d = _Draw(seed)

    def conv(name, cout, cin, k, gain=2.0):
        d.conv_w(name + ".weight", cout, cin, k, gain=gain)
        d.sd[name + ".bias"] = (0.05 * d.randn(cout)).float()

    def stage(n, inch, mid, outch, l):
        for i in range(n):
            b = f"{l}.{i}"
            conv(b + ".conv1", mid, inch, 1)
            conv(b + ".conv2", mid, mid, 3)
            conv(b + ".conv3", outch, mid, 1, gain=0.125)
            if inch != outch:
                conv(b + ".shortcut", outch, inch, 1, gain=1.0)
            inch = outch

    conv("backbone.stem.conv1", 64, 3, 7, gain=2.0 / 70.0 ** 2)
    inch, mid, outch = 64, 64, 256
    for s_, n in enumerate((3, 4, 6)):
        stage(n, inch, mid, outch, f"backbone.res{s_ + 2}")
        inch, mid, outch = outch, mid * 2, outch * 2
    conv("proposal_generator.rpn_head.conv", 1024, 1024, 3)
    conv("proposal_generator.rpn_head.objectness_logits", anchors, 1024, 1, gain=8.0)
    conv("proposal_generator.rpn_head.anchor_deltas", 4 * anchors, 1024, 1, gain=0.05)
    stage(3, 1024, 512, 2048, "roi_heads.res5")
    nc = num_classes
    d.sd["roi_heads.box_predictor.cls_score.weight"] = (d.randn(nc + 1, 2048) * 0.06).float()
    d.sd["roi_heads.box_predictor.cls_score.bias"] = (0.1 * d.randn(nc + 1)).float()
    d.sd["roi_heads.box_predictor.bbox_pred.weight"] = (d.randn(4 * nc, 2048) * 0.01).float()
    d.sd["roi_heads.box_predictor.bbox_pred.bias"] = (0.01 * d.randn(4 * nc)).float()
    stage(3, 1024, 512, 2048, "roi_heads.res5")   # the mask branch runs res5 again: second draw, first position (see _Draw)
    d.sd["roi_heads.mask_head.deconv.weight"] = (d.randn(2048, 256, 2, 2) * math.sqrt(2.0 / 2048)).float()
    d.sd["roi_heads.mask_head.deconv.bias"] = (0.05 * d.randn(256)).float()
    conv("roi_heads.mask_head.predictor", nc, 256, 1, gain=4.0)
    return d.state()


STATE = {"resnet50": resnet50_state, "retinaface_r50": retinaface_r50_state, "rcnn_r50c4": rcnn_r50c4_state}   # (yolov8n_state is added below its definition)


YOLOV5_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]  # yolov5s P3/P4/P5 (ultralytics yaml)


def yolov5_head_tensors(batch, classes=80, net_h=640, net_w=640, strides=(8, 16, 32), objects=(40, 160), seed=0, seg=False):
    """Planted-object inputs of the anchor-based YoloLayer: list of [B, 3*(5+classes(+32)), gh*gw] fp32 (channel-major, as the
    detect convolutions emit them, yolov5/src/model.cpp:331-343).  Objectness logits ~ N(-6, 1.5^2) with `objects` planted
    anchors per image (obj +2..+6, one class +3..+7, small clusters of neighbouring cells so that NMS has work)."""
    rng = np.random.default_rng(seed)
    info = 5 + classes + (32 if seg else 0)
    outs = []
    for s in strides:
        gh, gw = net_h // s, net_w // s
        x = rng.normal(0.0, 1.0, size=(batch, 3, info, gh * gw)).astype(np.float32)
        x[:, :, 4] = rng.normal(-6.0, 1.5, size=(batch, 3, gh * gw))
        x[:, :, 5:5 + classes] = rng.normal(-4.0, 1.5, size=(batch, 3, classes, gh * gw))
        outs.append(x)
    for b in range(batch):
        for _ in range(int(rng.integers(*objects))):
            l = int(rng.integers(0, len(strides)))
            gh, gw = net_h // strides[l], net_w // strides[l]
            cy, cx, k, c = int(rng.integers(0, gh)), int(rng.integers(0, gw)), int(rng.integers(0, 3)), int(rng.integers(0, classes))
            for dy, dx in ((0, 0), (0, 1), (1, 0), (1, 1)):
                if rng.uniform() < 0.35 and (dy, dx) != (0, 0):
                    continue
                e = min(cy + dy, gh - 1) * gw + min(cx + dx, gw - 1)
                outs[l][b, k, 4, e] = rng.uniform(2.0, 6.0)
                outs[l][b, k, 5 + c, e] = rng.uniform(3.0, 7.0)
                outs[l][b, k, 0:4, e] = rng.normal(0, 0.4, size=4)
    return [x.reshape(batch, 3 * info, -1) for x in outs]


STATE["yolov8n"] = yolov8n_state
