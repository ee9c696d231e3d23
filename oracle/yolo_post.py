"""ORACLE (test infrastructure): YOLOv8 decode + NMS restatements.

* ``decode_c`` / ``batch_nms_c``  — ctypes wrappers over oracle/csrc/yolo_post_ref.c (sequential C,
  follows yolov8/plugin/yololayer.cu:178-220,282-316 and yolov8/src/postprocess.cpp:71-129).
* ``decode_np``  — an independent vectorised NumPy restatement of the same decode
  (used to cross-check the C restatement).
* ``nms_py``     — an independent pure-Python restatement of nms() for small cases.

Parity status: parity unpinned (no reference goldens exist for this path).
"""
import ctypes
import math

import numpy as np

from . import lib

DET_FLOATS = 90  # yolov8/include/types.h:4-12 (4 + 1 + 1 + 32 + 17*3 + 1)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def decode_c(inputs, classes, net_h, net_w, strides, max_out=1000):
    """inputs: list of [B, 4+classes, gh*gw] float32 arrays (one per stride level)."""
    ins = [np.ascontiguousarray(x, dtype=np.float32) for x in inputs]
    batch = ins[0].shape[0]
    out = np.zeros((batch, 1 + max_out * DET_FLOATS), dtype=np.float32)
    ptrs = (ctypes.POINTER(ctypes.c_float) * len(ins))(*[_fp(x) for x in ins])
    st = (ctypes.c_int * len(strides))(*strides)
    lib().yolo_decode_ref(ptrs, batch, classes, net_h, net_w, st, len(strides), max_out, _fp(out))
    return out


def batch_nms_c(output, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    output = np.ascontiguousarray(output, dtype=np.float32)
    batch = output.shape[0]
    keep_idx = np.full((batch, max_out), -1, dtype=np.int32)
    keep_cnt = np.zeros((batch,), dtype=np.int32)
    keep_det = np.zeros((batch, max_out, 6), dtype=np.float32)
    lib().yolo_batch_nms_ref(
        _fp(output), batch, max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh),
        keep_idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
        keep_cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(keep_det))
    return keep_idx, keep_cnt, keep_det


def decode_np(inputs, classes, net_h, net_w, strides, max_out=1000):
    """Vectorised restatement of CalDetection (yololayer.cu:178-220), canonical (level, cell) order."""
    batch = inputs[0].shape[0]
    out = np.zeros((batch, 1 + max_out * DET_FLOATS), dtype=np.float32)
    one = np.float32(1.0)
    for b in range(batch):
        recs = []
        for x, s in zip(inputs, strides):
            gh, gw = net_h // s, net_w // s
            cur = np.asarray(x[b], dtype=np.float32)  # [4+classes, total]
            with np.errstate(over="ignore"):
                p = one / (one + np.exp(-cur[4:4 + classes], dtype=np.float32))
            # strict '>' scan starting from (0.0, class 0)  == first index attaining the maximum
            cls = np.argmax(p, axis=0)
            mx = p[cls, np.arange(p.shape[1])]
            cls = np.where(mx > 0, cls, 0)
            sel = np.nonzero(~(mx.astype(np.float64) < 0.1))[0]
            e = sel.astype(np.int64)
            row = (e // gw).astype(np.float32)
            col = (e % gw).astype(np.float32)
            half = np.float32(0.5)
            sf = np.float32(s)
            rec = np.zeros((len(e), DET_FLOATS), dtype=np.float32)
            rec[:, 0] = (col + half - cur[0, e]) * sf
            rec[:, 1] = (row + half - cur[1, e]) * sf
            rec[:, 2] = (col + half + cur[2, e]) * sf
            rec[:, 3] = (row + half + cur[3, e]) * sf
            rec[:, 4] = mx[e]
            rec[:, 5] = cls[e].astype(np.float32)
            recs.append(rec)
        rec = np.concatenate(recs, axis=0)[:max_out]
        out[b, 0] = len(rec)
        out[b, 1:1 + rec.size] = rec.reshape(-1)
    return out


def _iou(l, r):
    f = np.float32
    ib0, ib1 = max(l[0], r[0]), min(l[2], r[2])
    ib2, ib3 = max(l[1], r[1]), min(l[3], r[3])
    if ib2 > ib3 or ib0 > ib1:
        return f(0.0)
    inter = f(f(ib1 - ib0) * f(ib3 - ib2))
    uni = f(f(f(f(l[2] - l[0]) * f(l[3] - l[1])) + f(f(r[2] - r[0]) * f(r[3] - r[1]))) - inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return f(inter / uni)


def nms_py(output_row, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """Pure-Python restatement of nms() (postprocess.cpp:94-121) for one image; returns kept slot indices."""
    count = min(int(output_row[0]), max_out)
    buckets = {}
    for i in range(count):
        det = output_row[1 + DET_FLOATS * i: 1 + DET_FLOATS * (i + 1)]
        conf = det[4]
        if conf <= np.float32(conf_thresh) or math.isnan(conf):
            continue
        buckets.setdefault(float(det[5]), []).append((i, det))
    keep = []
    for cls in sorted(buckets):
        dets = sorted(buckets[cls], key=lambda t: (-float(t[1][4]), float(t[1][0]), t[0]))
        m = 0
        while m < len(dets):
            item = dets[m]
            keep.append(item[0])
            n = m + 1
            while n < len(dets):
                if _iou(item[1], dets[n][1]) > np.float32(nms_thresh):
                    del dets[n]
                else:
                    n += 1
            m += 1
    return keep


def gpu_postprocess_c(output, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """Mode "g" of the reference (yolov8/src/postprocess.cu:42-111), C restatement: [B, 1 + max_out*7]."""
    output = np.ascontiguousarray(output, dtype=np.float32)
    batch = output.shape[0]
    out = np.zeros((batch, 1 + max_out * 7), dtype=np.float32)
    lib().yolo_gpu_postprocess_ref(_fp(output), batch, max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh), _fp(out))
    return out


def gpu_postprocess_py(row, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """Independent pure-Python statement of the same mode for one image (cross-check of the C)."""
    f32 = np.float32
    count = min(int(row[0]), max_out)
    out = np.zeros(1 + max_out * 7, dtype=np.float32)
    out[0] = count
    det = np.asarray(row[1:1 + max_out * DET_FLOATS], dtype=np.float32).reshape(max_out, DET_FLOATS)
    for i in range(count):
        if not (det[i, 4] < conf_thresh):
            out[1 + 7 * i:1 + 7 * i + 6] = det[i, :6]
            out[1 + 7 * i + 6] = 1.0
    rec = out[1:].reshape(max_out, 7)

    def iou(a, b):
        cw = max(f32(min(a[2], b[2]) - max(a[0], b[0])), f32(0))
        ch = max(f32(min(a[3], b[3]) - max(a[1], b[1])), f32(0))
        c = f32(cw * ch)
        if c == 0:
            return f32(0)
        aa = f32(max(f32(a[2] - a[0]), f32(0)) * max(f32(a[3] - a[1]), f32(0)))
        bb = f32(max(f32(b[2] - b[0]), f32(0)) * max(f32(b[3] - b[1]), f32(0)))
        return f32(c / f32(f32(aa + bb) - c))

    keep = rec[:, 6].copy()
    for p in range(count):
        for i in range(count):
            if i == p or rec[p, 5] != rec[i, 5]:
                continue
            if rec[i, 4] >= rec[p, 4]:
                if rec[i, 4] == rec[p, 4] and i < p:
                    continue
                if iou(rec[p], rec[i]) > f32(nms_thresh):
                    keep[p] = 0.0
                    break
    rec[:, 6] = keep
    return out


# ---------------------------------------------------------------------------------------------------- YOLOv5 (anchor based)
DET5_FLOATS = 38  # yolov5/src/types.h:11-16


def v5_decode_c(inputs, classes, net_h, net_w, grids, anchors, max_out=1000, is_seg=False):
    """yolov5/plugin/yololayer.cu:161-227, C restatement.  inputs: [B, 3*(5+classes(+32)), gh*gw] per level; grids [(gw, gh)]."""
    ins = [np.ascontiguousarray(x, dtype=np.float32) for x in inputs]
    B, n = ins[0].shape[0], len(ins)
    out = np.zeros((B, 1 + max_out * DET5_FLOATS), dtype=np.float32)
    ptrs = (ctypes.POINTER(ctypes.c_float) * n)(*[_fp(x) for x in ins])
    gw = (ctypes.c_int * n)(*[g[0] for g in grids])
    gh = (ctypes.c_int * n)(*[g[1] for g in grids])
    an = np.ascontiguousarray(anchors, dtype=np.float32).reshape(n, 6)
    lib().yolov5_decode_ref(ptrs, n, B, classes, net_h, net_w, gw, gh, _fp(an), max_out, 1 if is_seg else 0, _fp(out))
    return out


def v5_batch_nms_c(output, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """yolov5/src/postprocess.cpp:30-80, C restatement -> keep_idx [B, max_out], keep_cnt [B], keep_det [B, max_out, 6]."""
    output = np.ascontiguousarray(output, dtype=np.float32)
    B = output.shape[0]
    keep_idx = np.full((B, max_out), -1, dtype=np.int32)
    keep_cnt = np.zeros((B,), dtype=np.int32)
    keep_det = np.zeros((B, max_out, 6), dtype=np.float32)
    L = lib()
    L.yolov5_nms_ref.restype = ctypes.c_int
    for b in range(B):
        keep_cnt[b] = L.yolov5_nms_ref(_fp(output[b]), max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh),
                                       keep_idx[b].ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(keep_det[b]))
    return keep_idx, keep_cnt, keep_det


# ---------------------------------------------------------------------------------------------------- YOLOv8 seg / pose / obb
def decode_ex_c(inputs, classes, net_h, net_w, strides, max_out=1000, nk=17, kpt_conf=0.0, seg=False, pose=False, obb=False):
    """yolov8/plugin/yololayer.cu:178-279 with the optional branches; inputs [B, 4+classes(+32)(+3nk)(+1), cells] per level."""
    ins = [np.ascontiguousarray(x, dtype=np.float32) for x in inputs]
    batch = ins[0].shape[0]
    out = np.zeros((batch, 1 + max_out * DET_FLOATS), dtype=np.float32)
    ptrs = (ctypes.POINTER(ctypes.c_float) * len(ins))(*[_fp(x) for x in ins])
    st = (ctypes.c_int * len(strides))(*strides)
    lib().yolo_decode_ex_ref(ptrs, batch, classes, net_h, net_w, st, len(strides), max_out, nk, ctypes.c_float(kpt_conf), int(seg), int(pose),
                             int(obb), _fp(out))
    return out


def batch_nms_obb_c(output, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """yolov8/src/postprocess.cpp:303-393 (ProbIoU) -> keep_idx [B, max_out], keep_cnt [B], keep_det [B, max_out, 7]."""
    output = np.ascontiguousarray(output, dtype=np.float32)
    B = output.shape[0]
    keep_idx = np.full((B, max_out), -1, dtype=np.int32)
    keep_cnt = np.zeros((B,), dtype=np.int32)
    keep_det = np.zeros((B, max_out, 7), dtype=np.float32)
    L = lib()
    L.yolo_nms_obb_ref.restype = ctypes.c_int
    for b in range(B):
        keep_cnt[b] = L.yolo_nms_obb_ref(_fp(output[b]), max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh),
                                         keep_idx[b].ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(keep_det[b]))
    return keep_idx, keep_cnt, keep_det


def gpu_postprocess_obb_c(output, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """cuda_decode_obb + cuda_nms_obb (yolov8/src/postprocess.cu), C restatement: [B, 1 + max_out*8]."""
    output = np.ascontiguousarray(output, dtype=np.float32)
    out = np.zeros((output.shape[0], 1 + max_out * 8), dtype=np.float32)
    lib().yolo_gpu_postprocess_obb_ref(_fp(output), output.shape[0], max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh), _fp(out))
    return out
