"""ORACLE (test infrastructure): CPU interpreter of a LOWERED kINT8 / kFP16 plan at the engine's own scales - VERDICT r4 "missing 3".

What tests/test_gpu_int8.py could say until round 5 was how far an int8 engine's detections sit from the fp32 oracle's - a figure that mixes two
things: what the CALIBRATOR chose (the reference's entropy search clips the tails the random-weight candidates live in) and whether the int8 KERNELS
compute what their scales say.  This interpreter removes the first: it takes the plan the engine runs - `trtx_plan_describe(lowered=True)`: the fused
ops, each tensor's storage / channel offset / dtype / calibrated scale, each convolution's int8 flags - and evaluates it on the CPU with the arithmetic
the kernels state:

  * an int8 tensor holds q = clamp(rint(x / s), -127, 127) and means q * s; an fp16 tensor holds half(x);
  * an int8 convolution (kernels/conv_igemm.hip, I8): weights per output channel w_q = clamp(rint(w * bn_scale / s_w)), s_w = max|w * bn_scale| / 127
    (runtime/pack.cpp conv_pack_weights_i8), integer products summed exactly (here: float64 over integer values), dequantised by the fp32 product
    s_in * s_w, + fp32 bias, activation, ONE fp16 rounding, (+ shortcut, rounded again), requantised with the fp32 reciprocal of the output scale;
  * an fp16 convolution: half(w * bn_scale), fp32 accumulation, the same epilogue; the stem rounds the image to fp16;
  * max-pools and the nearest resize move stored values (the int8 resize requantises by the ratio of the two scales, quant_ops.hip);
  * the fused detect tail (plugins/yolo_decode.hip yolo_head_score_kernel) = DFL softmax . arange + oracle/yolo_post.decode_c.

Calibration appears nowhere: the scales are read from the plan.  An int8 engine must reproduce THIS to within the few values a 1-ulp difference in an
activation moves across a rounding boundary - that is kernel parity independent of the calibrator (tests/test_gpu_int8.py).
Reference: yolov8/src/calibrator.cpp:9-74 and yolov8/src/model.cpp:317-324 (what sets kINT8 and feeds the scales), retinaface/retina_r50.cpp:12.
Op kinds handled: conv (stem / fp16 / int8), pool, pool_chain, resize, to_linear, yolo_head, plugin (Decode_TRT) - what the YOLOv8n and RetinaFace-R50
int8 plans lower to."""
import numpy as np
import torch
import torch.nn.functional as F

from . import det_post, yolo_post

L_CONV, L_SCALE, L_FC = 1, 5, 12
DT_F32, DT_F16, DT_I8 = 0, 1, 2


def _half(x):
    return x.half().float()


def _wvec(plan, ref):
    off, cnt = ref
    return None if cnt == 0 else torch.from_numpy(np.frombuffer(plan, dtype=np.float32, count=cnt, offset=off).copy())


def _act(y, code, alpha=0.1):
    # ACT_* of kernels.h; SiLU / sigmoid as the epilogues compute them (x * 1 / (1 + exp(-x)) in fp32)
    if code == 0:
        return y
    if code == 1:
        return torch.relu(y)
    if code == 2:
        return torch.sigmoid(y)
    if code == 3:
        return y * torch.sigmoid(y)
    if code == 4:
        return F.leaky_relu(y, alpha)
    if code == 5:
        return torch.tanh(y)
    raise NotImplementedError(f"activation {code}")


def run(plan, net, low, inputs, batch):
    """plan: serialized plan bytes; net / low: trtx_plan_describe(lowered=False / True); inputs: {binding name: array [B, ...]}.
    Returns {output binding name: np.ndarray}."""
    layers_by_name = {}
    for l in net["layers"]:
        layers_by_name.setdefault(l["name"], l)
    consumers = {}
    for l in net["layers"]:
        for t in l["inputs"]:
            consumers.setdefault(t, []).append(l)
    tens = low["tensors"]
    store = {}    # storage id -> torch float32 [B, H, W, ld] (NHWC) or flat [B, n] (LINEAR): the VALUES the stored bits mean

    def geom(t):
        d = t["dims"]
        return d[-3], d[-2], d[-1]

    def view(tid):
        t = tens[tid]
        C, H, W = geom(t)
        s = store[t["storage"]]
        return s[..., t["coff"]:t["coff"] + C]

    def alloc(tid):
        t = tens[tid]
        if t["storage"] in store:
            return
        if t["layout"] == "nhwc":
            _, H, W = geom(t)
            store[t["storage"]] = torch.zeros(batch, H, W, t["ld"])
        else:
            store[t["storage"]] = None

    def put(tid, v):
        """v: real values [B, H, W, C] -> stored with the tensor's dtype"""
        t = tens[tid]
        alloc(tid)
        if t["dtype"] == DT_I8:
            inv = np.float32(1.0) / np.float32(t["scale"])
            q = torch.clamp(torch.round(_half(v) * float(inv)), -127, 127)
            v = q * float(np.float32(t["scale"]))
        elif t["dtype"] == DT_F16:
            v = _half(v)
        C = geom(t)[0]
        store[t["storage"]][..., t["coff"]:t["coff"] + C] = v

    # network inputs: LINEAR fp32 bindings
    name_of_net = {t["id"]: t["name"] for t in net["tensors"]}
    for t in tens:
        if t["layout"] == "linear" and t["net"] >= 0 and name_of_net.get(t["net"]) in inputs and t["storage"] not in store:
            store[t["storage"]] = torch.as_tensor(np.asarray(inputs[name_of_net[t["net"]]], dtype=np.float32))
    out = {}
    ops = []
    for op in low["ops"]:   # a grouped launch is its member convolutions, each exactly as its own launch
        ops.extend([dict(m, kind="conv") for m in op["members"]] if op["kind"] == "conv_group" else [op])
    for op in ops:
        kind = op["kind"]
        if kind == "conv":
            assert not op.get("up_c"), "folded upsample: lower with TRTX_FOLD_UPSAMPLE=0 (kINT8 plans never fold)"
            layer = layers_by_name[op["name"].split(" [")[0]]
            assert layer["kind"] in (L_CONV, L_FC), op["name"]
            w0, w1 = _wvec(plan, layer["w"][0]), _wvec(plan, layer["w"][1])
            cout = op["cout"]
            tin, tout = tens[op["in"][0]], tens[op["out"][0]]
            if op["stem"]:
                x = store[tin["storage"]].reshape(batch, *tin["dims"][-3:])            # [B, C, H, W] fp32 image
                cin = x.shape[1]
            else:
                x = view(op["in"][0]).permute(0, 3, 1, 2)
                cin = x.shape[1]
            w = w0.reshape(cout, cin, *op["k"])
            sc = torch.ones(cout)
            bias = w1.clone() if w1 is not None else torch.zeros(cout)
            if op["bn_folded"]:   # the IScaleLayer the lowering folded: the convolution output's consumer
                sl = [c for c in consumers[layer["outputs"][0]] if c["kind"] == L_SCALE][0]
                shift, scale = _wvec(plan, sl["w"][0]), _wvec(plan, sl["w"][1])
                sc = scale if scale is not None else sc
                bias = bias * sc + (shift if shift is not None else 0.0)
            wf = (w * sc[:, None, None, None]).float()                                  # fp32 product, as pack_weights forms it
            stride, pad = op["stride"], layer["padding"]
            if op["i8"][0]:
                s_in = np.float32(tin["scale"])
                xq = torch.round(x / float(s_in))                                       # the stored integers
                amax = wf.abs().amax(dim=(1, 2, 3))
                s_w = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax)).float()
                wq = torch.clamp(torch.round(wf / s_w[:, None, None, None]), -127, 127)
                acc = F.conv2d(xq.double(), wq.double(), None, stride=stride, padding=pad)   # exact integers
                cscale = (s_w * float(s_in)).float()
                y = acc.float() * cscale[None, :, None, None] + bias[None, :, None, None]
            elif op["igemm"] or op["stem"]:
                xh = _half(x)
                wh = _half(wf)
                y = F.conv2d(xh.double(), wh.double(), None, stride=stride, padding=pad).float() + bias[None, :, None, None]
            else:   # the direct kernel (K < 32, grouped, dilated ...: nhwc_ops.hip conv_direct_kernel): fp32 weights, fp16 activations
                y = F.conv2d(_half(x).double(), wf.double(), None, stride=stride, padding=pad, groups=layer["groups"], dilation=layer["dilation"]).float() + bias[None, :, None, None]
            y = _half(_act(y, op["act1"], op.get("alpha1", 0.1)))
            if op["residual"]:
                r = view(op["in"][1]).permute(0, 3, 1, 2)
                y = _half(_act(y + _half(r), op["act2"], op.get("alpha2", 0.1)))
            elif op["act2"]:
                y = _half(_act(y, op["act2"], op.get("alpha2", 0.1)))
            put(op["out"][0], y.permute(0, 2, 3, 1))
        elif kind in ("pool", "pool_chain"):
            layer = layers_by_name[op["name"].split(" [")[0]]
            x = view(op["in"][0]).permute(0, 3, 1, 2)
            k, s_, p_ = layer["kernel"], layer["stride"], layer["padding"]
            cur = x
            for o in op["out"]:
                if layer["op"] == 0:
                    cur = F.max_pool2d(cur, k, s_, p_)
                else:
                    cur = F.avg_pool2d(cur, k, s_, p_, count_include_pad=not layer["avg_exclusive"])
                put(o, cur.permute(0, 2, 3, 1))
        elif kind == "resize":
            tin, tout = tens[op["in"][0]], tens[op["out"][0]]
            x = view(op["in"][0]).permute(0, 3, 1, 2)
            H, W = geom(tout)[1:]
            if tin["dtype"] == DT_I8:   # quant_ops.hip: q_out = clamp(rint(q_in * (s_in / s_out)))
                q = torch.round(x / float(np.float32(tin["scale"])))
                ratio = np.float32(tin["scale"]) / np.float32(tout["scale"])
                q = torch.clamp(torch.round(F.interpolate(q, size=(H, W), mode="nearest") * float(ratio)), -127, 127)
                alloc(op["out"][0])
                store[tout["storage"]][..., tout["coff"]:tout["coff"] + geom(tout)[0]] = (q * float(np.float32(tout["scale"]))).permute(0, 2, 3, 1)
            else:
                put(op["out"][0], F.interpolate(x, size=(H, W), mode="nearest").permute(0, 2, 3, 1))
        elif kind == "to_nhwc":
            t = tens[op["in"][0]]
            C, H, W = geom(tens[op["out"][0]])
            put(op["out"][0], store[t["storage"]].reshape(batch, C, H, W).permute(0, 2, 3, 1))
        elif kind == "to_linear":
            t = tens[op["out"][0]]
            store[t["storage"]] = view(op["in"][0]).permute(0, 3, 1, 2).contiguous().reshape(batch, -1)
        elif kind == "yolo_head":
            layer = layers_by_name[op["name"].split(" [")[0]]
            blob = bytes.fromhex(layer["plugin_blob"])
            hdr = np.frombuffer(blob, dtype=np.int32, count=8)
            classes, net_w, net_h, max_out, ns = int(hdr[0]), int(hdr[4]), int(hdr[5]), int(hdr[6]), int(hdr[7])
            strides = [int(v) for v in np.frombuffer(blob, dtype=np.int32, count=ns, offset=32)]
            arrs = []
            ar = torch.arange(16, dtype=torch.float32)
            for tid in op["in"]:
                h = view(tid)                                   # [B, gh, gw, 64 + classes]
                B, gh, gw, _ = h.shape
                box = torch.softmax(h[..., :64].reshape(B, gh * gw, 4, 16), dim=-1) @ ar     # DFL: expectation over the 16 bins (block.cpp:239-257)
                cls = h[..., 64:64 + classes].reshape(B, gh * gw, classes)
                arrs.append(np.ascontiguousarray(torch.cat([box, cls], -1).permute(0, 2, 1).numpy()))
            dec = yolo_post.decode_c(arrs, classes, net_h, net_w, strides, max_out)
            out[name_of_net[tens[op["out"][0]]["net"]]] = dec
        elif kind == "plugin":
            layer = layers_by_name[op["name"].split(" [")[0]]
            assert layer["plugin_type"] == "Decode_TRT", layer["plugin_type"]
            net_h, net_w = (int(v) for v in np.frombuffer(bytes.fromhex(layer["plugin_blob"]), dtype=np.int32, count=2))
            arrs = [np.ascontiguousarray(store[tens[t]["storage"]].numpy().reshape(batch, 32, -1)) for t in op["in"]]
            out[name_of_net[tens[op["out"][0]]["net"]]] = det_post.retina_decode(arrs, net_h, net_w)
        else:
            raise NotImplementedError(f"lowered op kind {kind}")
    # marked outputs that are plain tensors (ad-hoc test networks): the LINEAR fp32 copy the engine writes into the binding
    for t in tens:
        nm = name_of_net.get(t["net"])
        if t["layout"] == "linear" and nm and nm not in out and nm not in inputs and store.get(t["storage"]) is not None and any(nt["name"] == nm and nt["is_output"] for nt in net["tensors"]):
            out[nm] = store[t["storage"]].numpy()
    return out
