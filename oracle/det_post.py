"""ORACLE (test infrastructure): ctypes wrappers over oracle/csrc/retina_rcnn_ref.c — sequential C restatements
of RetinaFace decode/NMS (retinaface/decode.cu:110-191, common.hpp:91-130) and of the R-CNN plugin chain
(rcnn/RpnDecode.cu, RpnNms.cu, RoiAlign.cu, PredictorDecode.cu, BatchedNms.cu).  Parity unpinned."""
import ctypes

import numpy as np

from . import lib

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def _fp(a):
    return a.ctypes.data_as(_f)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def retina_out_floats(net_h, net_w):
    return 1 + sum((net_h // s) * (net_w // s) * 2 * 15 for s in (8, 16, 32))


def retina_decode(inputs, net_h, net_w):
    ins = [_c(x) for x in inputs]
    B = ins[0].shape[0]
    out = np.zeros((B, retina_out_floats(net_h, net_w)), np.float32)
    ptrs = (_f * 3)(*[_fp(x) for x in ins])
    lib().retina_decode_ref(ptrs, B, net_h, net_w, _fp(out))
    return out


def retina_nms(decoded, conf_thresh=0.1, nms_thresh=0.4, max_keep=1000):
    d = _c(decoded)
    B = d.shape[0]
    L = lib()
    L.retina_nms_ref.restype = ctypes.c_int
    idx = np.full((B, max_keep), -1, np.int32)
    cnt = np.zeros((B,), np.int32)
    for b in range(B):
        cnt[b] = L.retina_nms_ref(_fp(d[b]), ctypes.c_double(conf_thresh), ctypes.c_float(nms_thresh), max_keep,
                                  idx[b].ctypes.data_as(_i))
    return idx, cnt


def generate_anchors(sizes=(32, 64, 128, 256, 512), ratios=(0.5, 1.0, 2.0)):
    """rcnn/rcnn.cpp:62-77 (float arithmetic)."""
    res = []
    for a in sizes:
        area = np.float32(a) * np.float32(a)
        for ar in ratios:
            w = np.float32(np.sqrt(np.float32(area / np.float32(ar))))
            h = np.float32(np.float32(ar) * w)
            res += [-w / 2.0, -h / 2.0, w / 2.0, h / 2.0]
    return np.asarray(res, np.float32)


def rpn_decode(scores, deltas, h, w, img_h, img_w, stride, anchors, top_n):
    s, d, a = _c(scores), _c(deltas), _c(anchors)
    B = s.shape[0]
    os_, ob = np.zeros((B, top_n), np.float32), np.zeros((B, top_n, 4), np.float32)
    lib().rpn_decode_ref(B, _fp(s), _fp(d), h, w, img_h, img_w, ctypes.c_float(stride), _fp(a), a.size // 4, top_n,
                         _fp(os_), _fp(ob))
    return os_, ob


def rpn_nms(scores, boxes, post, thresh):
    s, b = _c(scores), _c(boxes)
    B, pre = s.shape
    out = np.zeros((B, post, 4), np.float32)
    lib().rpn_nms_ref(B, _fp(s), _fp(b), pre, post, ctypes.c_float(thresh), _fp(out))
    return out


def roi_align(boxes, feats, res, scale, sampling_ratio=0):
    b, f = _c(boxes), _c(feats)
    B, P = b.shape[:2]
    C, fh, fw = f.shape[1:]
    out = np.zeros((B, P, C, res, res), np.float32)
    lib().roi_align_ref(B, _fp(b), _fp(f), res, ctypes.c_float(scale), sampling_ratio, P, C, fh, fw, _fp(out))
    return out


def predictor_decode(scores, deltas, proposals, img_h, img_w, weights=(10.0, 10.0, 5.0, 5.0)):
    s, d, p = _c(scores), _c(deltas), _c(proposals)
    B, N, C = s.shape
    w = _c(weights)
    os_, ob, oc = np.zeros((B, N), np.float32), np.zeros((B, N, 4), np.float32), np.zeros((B, N), np.float32)
    lib().predictor_decode_ref(B, _fp(s), _fp(d), _fp(p), N, C, img_h, img_w, _fp(w), _fp(os_), _fp(ob), _fp(oc))
    return os_, ob, oc


def batched_nms(method, scores, boxes, classes, dets, thresh):
    s, b, c = _c(scores), _c(boxes), _c(classes)
    B, count = s.shape
    os_, ob, oc = np.zeros((B, dets), np.float32), np.zeros((B, dets, 4), np.float32), np.zeros((B, dets), np.float32)
    lib().batched_nms_ref(method, B, _fp(s), _fp(b), _fp(c), count, dets, ctypes.c_float(thresh), _fp(os_), _fp(ob),
                          _fp(oc))
    return os_, ob, oc


def mask_select(labels, masks):
    """rcnn/MaskRcnnInference.cu:8-30: out[b, d, 0] = sigmoid(masks[b, d, class(d)]); class ids outside [0, C) give zeros
    (the reference leaves those planes unwritten).  NumPy (the arithmetic is one sigmoid)."""
    lab = np.asarray(labels, np.float32)
    m = np.asarray(masks, np.float32)
    B, D, C = m.shape[:3]
    out = np.zeros((B, D, 1) + m.shape[3:], np.float32)
    for b in range(B):
        for d in range(D):
            c = int(lab[b, d])
            if 0 <= c < C:
                out[b, d, 0] = (np.float32(1.0) / (np.float32(1.0) + np.exp(-m[b, d, c], dtype=np.float32))).astype(np.float32)
    return out
