"""ctypes wrappers of the RetinaFace / R-CNN plugin operators (include/trtx_hip.h, section 1)."""
import ctypes

from .capi import _p, _stream, check, lib


def _ws(nbytes, dev):
    import torch
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def _L():
    L = lib()
    for f in ("trtx_retina_decode_output_floats", "trtx_retina_decode_workspace", "trtx_retina_nms_workspace",
              "trtx_rpn_decode_workspace", "trtx_sorted_nms_workspace", "trtx_predictor_decode_workspace"):
        getattr(L, f).restype = ctypes.c_size_t
    return L


def retina_decode(inputs, net_h, net_w):
    import torch
    L = _L()
    ins = [x.contiguous() for x in inputs]
    B, dev = ins[0].shape[0], ins[0].device
    out = torch.empty((B, L.trtx_retina_decode_output_floats(net_h, net_w)), dtype=torch.float32, device=dev)
    wsb = L.trtx_retina_decode_workspace(B, net_h, net_w)
    ws = _ws(wsb, dev)
    arr = (ctypes.c_void_p * 3)(*[x.data_ptr() for x in ins])
    check(L.trtx_retina_decode(arr, B, net_h, net_w, _p(out), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), "trtx_retina_decode")
    return out


def retina_nms(dec, net_h, net_w, conf_thresh=0.1, nms_thresh=0.4, max_keep=1000):
    import torch
    L = _L()
    B, dev = dec.shape[0], dec.device
    ws = _ws(L.trtx_retina_nms_workspace(B, net_h, net_w), dev)
    idx = torch.full((B, max_keep), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    det = torch.zeros((B, max_keep, 15), dtype=torch.float32, device=dev)
    check(L.trtx_retina_nms(_p(dec), B, net_h, net_w, ctypes.c_double(conf_thresh), ctypes.c_float(nms_thresh), max_keep,
                            _p(idx), _p(cnt), _p(det), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), "trtx_retina_nms")
    return idx, cnt, det


def rpn_decode(scores, deltas, h, w, img_h, img_w, stride, anchors_np, top_n):
    import numpy as np
    import torch
    L = _L()
    B, dev = scores.shape[0], scores.device
    A = anchors_np.size // 4
    a = np.ascontiguousarray(anchors_np, dtype=np.float32)
    ws = _ws(L.trtx_rpn_decode_workspace(B, A, h, w), dev)
    os_ = torch.empty((B, top_n), dtype=torch.float32, device=dev)
    ob = torch.empty((B, top_n, 4), dtype=torch.float32, device=dev)
    check(L.trtx_rpn_decode(B, _p(scores), _p(deltas), h, w, img_h, img_w, ctypes.c_float(stride), a.ctypes.data_as(ctypes.c_void_p),
                            A, top_n, _p(os_), _p(ob), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), "trtx_rpn_decode")
    return os_, ob


def rpn_nms(scores, boxes, post, thresh):
    import torch
    L = _L()
    B, pre = scores.shape
    dev = scores.device
    ws = _ws(L.trtx_sorted_nms_workspace(B, pre), dev)
    out = torch.empty((B, post, 4), dtype=torch.float32, device=dev)
    check(L.trtx_rpn_nms(B, _p(scores), _p(boxes), pre, post, ctypes.c_float(thresh), _p(out), _p(ws), ctypes.c_size_t(ws.numel()),
                         _stream()), "trtx_rpn_nms")
    return out


def roi_align(boxes, feats, res, scale, sampling_ratio=0):
    import torch
    L = _L()
    B, P = boxes.shape[:2]
    C, fh, fw = feats.shape[1:]
    out = torch.empty((B, P, C, res, res), dtype=torch.float32, device=boxes.device)
    check(L.trtx_roi_align(B, _p(boxes), _p(feats), res, ctypes.c_float(scale), sampling_ratio, P, C, fh, fw, _p(out), _stream()),
          "trtx_roi_align")
    return out


def roi_align_nhwc_f16(boxes, feats_nhwc, res, scale, sampling_ratio=0):
    """Engine-native RoIAlign: feats_nhwc CUDA fp16 [B, fh, fw, C] -> fp16 [B*P, res, res, C]."""
    import torch
    L = _L()
    B, P = boxes.shape[:2]
    fh, fw, C = feats_nhwc.shape[1:]
    out = torch.empty((B * P, res, res, C), dtype=torch.float16, device=boxes.device)
    check(L.trtx_roi_align_nhwc_f16(B, _p(boxes), _p(feats_nhwc), feats_nhwc.stride(2), res, ctypes.c_float(scale), sampling_ratio, P, C, fh, fw,
                                    _p(out), C, _stream()), "trtx_roi_align_nhwc_f16")
    return out


def predictor_decode(scores, deltas, proposals, img_h, img_w, weights=(10.0, 10.0, 5.0, 5.0)):
    import numpy as np
    import torch
    L = _L()
    B, N, C = scores.shape
    dev = scores.device
    w = np.asarray(weights, dtype=np.float32)
    ws = _ws(L.trtx_predictor_decode_workspace(B, N, C), dev)
    os_ = torch.empty((B, N), dtype=torch.float32, device=dev)
    ob = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    oc = torch.empty((B, N), dtype=torch.float32, device=dev)
    check(L.trtx_predictor_decode(B, _p(scores), _p(deltas), _p(proposals), N, C, img_h, img_w, w.ctypes.data_as(ctypes.c_void_p),
                                  _p(os_), _p(ob), _p(oc), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), "trtx_predictor_decode")
    return os_, ob, oc


def batched_nms(method, scores, boxes, classes, dets, thresh):
    import torch
    L = _L()
    B, count = scores.shape
    dev = scores.device
    ws = _ws(L.trtx_sorted_nms_workspace(B, count), dev)
    os_ = torch.empty((B, dets), dtype=torch.float32, device=dev)
    ob = torch.empty((B, dets, 4), dtype=torch.float32, device=dev)
    oc = torch.empty((B, dets), dtype=torch.float32, device=dev)
    check(L.trtx_batched_nms(method, B, _p(scores), _p(boxes), _p(classes), count, dets, ctypes.c_float(thresh), _p(os_), _p(ob),
                             _p(oc), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), "trtx_batched_nms")
    return os_, ob, oc


def mask_rcnn_inference(labels, masks):
    """maskRcnnInference (rcnn/MaskRcnnInference.cu:8-62). labels [B, D] class ids as floats, masks [B, D, C, S, S]
    -> [B, D, 1, S, S]."""
    import torch
    B, D, C, S, _ = masks.shape
    out = torch.empty((B, D, 1, S, S), dtype=torch.float32, device=masks.device)
    check(_L().trtx_mask_rcnn_inference(B, _p(labels.contiguous()), _p(masks.contiguous()), D, S, C, _p(out), _stream()),
          "trtx_mask_rcnn_inference")
    return out
