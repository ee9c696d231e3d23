"""ORACLE (test infrastructure): layer-by-layer PyTorch-CPU fp32 interpreter of a serialized network
definition (the un-fused graph the host builder recorded, as exported by `trtx_plan_describe`).

It gives the TensorRT-layer semantics the reference builders rely on an executable,
independent statement: the GPU engine runs the *lowered, fused* plan, this runs the *original* layer list,
so a match checks both the fusion/lowering passes and the kernels.  Plugins are evaluated with the NumPy/C
restatements in this package.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import det_post, yolo_post

(L_INPUT, L_CONV, L_DECONV, L_ACTIVATION, L_POOLING, L_SCALE, L_ELEMENTWISE, L_CONCAT, L_SLICE, L_SHUFFLE, L_RESIZE,
 L_SOFTMAX, L_FC, L_MATMUL, L_CONSTANT, L_REDUCE, L_PLUGIN, L_IDENTITY) = range(18)


def _w(plan, ref):
    off, cnt = ref
    if cnt == 0:
        return None
    return torch.from_numpy(np.frombuffer(plan, dtype=np.float32, count=cnt, offset=off).copy())


def run(desc, plan, inputs, batch=None, keep=None):
    """desc: dict from trtx_plan_describe(lowered=False); plan: bytes; inputs: {name: np/torch array}.
    Implicit-batch networks take inputs shaped [B, *dims].  Returns {output name: torch tensor}
    (plus every tensor id listed in `keep`)."""
    explicit = desc["explicit_batch"]
    T = {}
    for t in desc["tensors"]:
        if t["is_input"]:
            v = torch.as_tensor(np.asarray(inputs[t["name"]], dtype=np.float32))
            if not explicit:
                assert list(v.shape[1:]) == t["dims"], (v.shape, t["dims"])
                batch = v.shape[0]
            T[t["id"]] = v
    b0 = 0 if explicit else 1  # index of the first logical dim in the torch tensors

    def lead(x):  # view with a leading batch dim in both modes
        return x if not explicit else x

    for l in desc["layers"]:
        k = l["kind"]
        ins = [T[i] for i in l["inputs"]]
        w0, w1, w2 = (_w(plan, r) for r in l["w"])
        if k in (L_CONV, L_DECONV, L_FC):
            x = ins[0]
            nd = x.dim()
            x4 = x.reshape(-1, *x.shape[-3:]) if nd != 4 else x
            cin = x4.shape[1]
            if k == L_FC:
                y = F.linear(x4.flatten(1), w0.reshape(l["nb_out"], -1), w1).reshape(x4.shape[0], l["nb_out"], 1, 1)
            elif k == L_CONV:
                wt = w0.reshape(l["nb_out"], cin // l["groups"], *l["kernel"])
                y = F.conv2d(x4, wt, w1, stride=l["stride"], padding=l["padding"], dilation=l["dilation"], groups=l["groups"])
            else:
                wt = w0.reshape(cin, l["nb_out"] // l["groups"], *l["kernel"])
                y = F.conv_transpose2d(x4, wt, w1, stride=l["stride"], padding=l["padding"], dilation=l["dilation"],
                                       groups=l["groups"])
            out = y.reshape(*x.shape[:-3], *y.shape[1:]) if nd != 4 else y
        elif k == L_ACTIVATION:
            x = ins[0]
            out = {0: torch.relu, 1: torch.sigmoid, 2: torch.tanh, 3: lambda t: F.leaky_relu(t, l["alpha"])}[l["op"]](x)
        elif k == L_POOLING:
            x = ins[0]
            if l["op"] == 0:
                out = F.max_pool2d(x, l["kernel"], l["stride"], l["padding"])
            else:
                out = F.avg_pool2d(x, l["kernel"], l["stride"], l["padding"], count_include_pad=not l["avg_exclusive"])
        elif k == L_SCALE:
            x = ins[0]
            ca = 1 if (explicit and x.dim() >= 4) else max(x.dim() - b0 - 3, 0) + b0
            shape = [1] * x.dim()
            if l["op"] == 1:
                shape[ca] = x.shape[ca]
            sh = w0.reshape(shape) if w0 is not None else 0.0
            sc = w1.reshape(shape) if w1 is not None else 1.0
            out = x * sc + sh
            if w2 is not None and not torch.all(w2 == 1):
                out = out ** w2.reshape(shape)
        elif k == L_ELEMENTWISE:
            a, b = ins
            if not explicit:  # a constant (no batch dim) broadcasts against a batched tensor
                if a.dim() == b.dim() - 1:
                    a = a.unsqueeze(0)
                if b.dim() == a.dim() - 1:
                    b = b.unsqueeze(0)
            out = {0: torch.add, 1: torch.mul, 2: torch.maximum, 3: torch.minimum, 4: torch.sub, 5: torch.div,
                   6: torch.pow}[l["op"]](a, b)
        elif k == L_CONCAT:
            out = torch.cat(ins, dim=l["axis"] + b0)
        elif k == L_SLICE:
            x = ins[0]
            idx = [slice(None)] * b0
            for s, n, st in zip(l["start"], l["size"], l["step"]):
                idx.append(slice(s, s + (n - 1) * st + 1, st))
            out = x[tuple(idx)]
        elif k == L_SHUFFLE:
            x = ins[0]
            nd = x.dim() - b0
            p1 = l["perm1"][:nd]
            x = x.permute(*range(b0), *[p + b0 for p in p1])
            if l["has_reshape"]:
                r = list(l["reshape"])
                for i, v in enumerate(r):
                    if v == 0:
                        r[i] = x.shape[b0 + i]
                x = x.reshape(*x.shape[:b0], *r)
            nd2 = x.dim() - b0
            p2 = l["perm2"][:nd2]
            out = x.permute(*range(b0), *[p + b0 for p in p2]).contiguous()
        elif k == L_RESIZE:
            x = ins[0]
            od = desc["tensors"][l["outputs"][0]]["dims"]
            assert l["op"] == 0
            out = F.interpolate(x, size=od[-2:], mode="nearest")
        elif k == L_SOFTMAX:
            x = ins[0]
            nd = x.dim() - b0
            if l["axis"] < 0:
                ax = max(0, nd - 3)
            else:
                ax = [i for i in range(nd) if (l["axis"] >> i) & 1][0]
            out = F.softmax(x, dim=ax + b0)
        elif k == L_MATMUL:
            a, b = ins
            if l["mm_op"][0] == 1:
                a = a.transpose(-1, -2)
            if l["mm_op"][1] == 1:
                b = b.transpose(-1, -2)
            out = torch.matmul(a, b)
        elif k == L_CONSTANT:
            out = w0.reshape(l["out_dims"])
        elif k == L_REDUCE:
            x = ins[0]
            nd = x.dim() - b0
            axes = [i + b0 for i in range(nd) if (l["axis"] >> i) & 1]
            keep_d = bool(l["keep_dims"])
            out = {0: lambda: x.sum(axes, keepdim=keep_d), 4: lambda: x.mean(axes, keepdim=keep_d),
                   2: lambda: x.amax(axes, keepdim=keep_d)}[l["op"]]()
        elif k == L_IDENTITY:
            out = ins[0]
        elif k == L_PLUGIN:
            outs = _plugin(l, ins, batch)
            for oid, o in zip(l["outputs"], outs):
                T[oid] = o
            continue
        else:
            raise NotImplementedError(f"layer kind {k}")
        od = desc["tensors"][l["outputs"][0]]["dims"]
        got = list(out.shape[b0:]) if (k != L_CONSTANT) else list(out.shape)
        assert got == od, (l["name"], got, od)
        T[l["outputs"][0]] = out
    res = {t["name"]: T[t["id"]] for t in desc["tensors"] if t["is_output"]}
    if keep:
        for i in keep:
            res[i] = T[i]
    return res


def _plugin(l, ins, batch):
    blob = bytes.fromhex(l["plugin_blob"])
    if l["plugin_type"] == "YoloLayer_TRT":
        # serialization layout: yolov8/plugin/yololayer.cu:75-101
        hdr = np.frombuffer(blob, dtype=np.int32, count=8)
        classes, net_w, net_h, max_out, ns = int(hdr[0]), int(hdr[4]), int(hdr[5]), int(hdr[6]), int(hdr[7])
        strides = [int(v) for v in np.frombuffer(blob, dtype=np.int32, count=ns, offset=32)]
        arrs = [np.ascontiguousarray(t.numpy().reshape(batch, 4 + classes, -1)) for t in ins]
        out = yolo_post.decode_c(arrs, classes, net_h, net_w, strides, max_out)
        return [torch.from_numpy(out).reshape(batch, -1, 1, 1)]
    if l["plugin_type"] == "Decode_TRT":
        # blob: int net_h, int net_w (our extension of the reference's empty blob, decode.cu:19-26)
        net_h, net_w = (int(v) for v in np.frombuffer(blob, dtype=np.int32, count=2))
        arrs = [np.ascontiguousarray(t.numpy().reshape(batch, 32, -1)) for t in ins]
        out = det_post.retina_decode(arrs, net_h, net_w)
        return [torch.from_numpy(out).reshape(batch, -1, 1, 1)]
    # R-CNN plugins: the reference's own blob layouts (rcnn/*Plugin.h deserialize()), size_t = u64
    u64 = lambda off, n=1: [int(v) for v in np.frombuffer(blob, dtype=np.uint64, count=n, offset=off)]  # noqa: E731
    i32 = lambda off, n=1: [int(v) for v in np.frombuffer(blob, dtype=np.int32, count=n, offset=off)]  # noqa: E731
    f32 = lambda off, n=1: [float(v) for v in np.frombuffer(blob, dtype=np.float32, count=n, offset=off)]  # noqa: E731
    arr = [np.ascontiguousarray(t.numpy()) for t in ins]
    if l["plugin_type"] == "RpnDecode":
        top_n, na = i32(0)[0], u64(4)[0]
        anchors = np.frombuffer(blob, dtype=np.float32, count=na, offset=12).copy()
        o = 12 + 4 * na
        stride = f32(o)[0]
        fh, fw, ih, iw = u64(o + 4, 4)
        s, b = det_post.rpn_decode(arr[0], arr[1], fh, fw, ih, iw, stride, anchors, top_n)
        return [torch.from_numpy(s).reshape(batch, top_n, 1), torch.from_numpy(b)]
    if l["plugin_type"] == "RpnNms":
        thresh = f32(0)[0]
        post, pre = i32(4)[0], u64(8)[0]
        return [torch.from_numpy(det_post.rpn_nms(arr[0].reshape(batch, pre), arr[1], post, thresh))]
    if l["plugin_type"] == "RoiAlign":
        res = i32(0)[0]
        scale = f32(4)[0]
        sampling = i32(8)[0]
        return [torch.from_numpy(det_post.roi_align(arr[0], arr[1], res, scale, sampling))]
    if l["plugin_type"] == "PredictorDecode":
        n, c, ih, iw = i32(0, 4)
        w = f32(24, u64(16)[0])
        s, b, cl = det_post.predictor_decode(arr[0].reshape(batch, n, c), arr[1].reshape(batch, n, 4 * c), arr[2], ih, iw, w)
        return [torch.from_numpy(s).reshape(batch, n, 1), torch.from_numpy(b), torch.from_numpy(cl).reshape(batch, n, 1)]
    if l["plugin_type"] == "BatchedNms":
        method = i32(0)[0]
        thresh = f32(4)[0]
        dets, count = i32(8)[0], u64(12)[0]
        s, b, cl = det_post.batched_nms(method, arr[0].reshape(batch, count), arr[1], arr[2].reshape(batch, count), dets, thresh)
        return [torch.from_numpy(s).reshape(batch, dets, 1), torch.from_numpy(b), torch.from_numpy(cl).reshape(batch, dets, 1)]
    if l["plugin_type"] == "MaskRcnnInference":
        dets, size, classes = i32(0, 3)
        out = det_post.mask_select(arr[0].reshape(batch, dets), arr[1].reshape(batch, dets, classes, size, size))
        return [torch.from_numpy(out)]
    raise NotImplementedError(l["plugin_type"])
