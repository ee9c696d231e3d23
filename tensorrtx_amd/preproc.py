"""ctypes wrappers of the letterbox pre-processing entry points (include/trtx_hip.h; reference yolov8/src/preprocess.cu)."""
import ctypes

import numpy as np

from .capi import _p, _stream, check, lib


def letterbox_matrix(src_w, src_h, dst_w, dst_h):
    m = np.zeros(6, np.float32)
    lib().trtx_letterbox_matrix(src_w, src_h, dst_w, dst_h, m.ctypes.data_as(ctypes.c_void_p))
    return m


def letterbox_batch(images_dev, dst_w, dst_h, out=None):
    """images_dev: list of CUDA uint8 tensors [H, W, 3] (BGR).  -> CUDA fp32 [B, 3, dst_h, dst_w]"""
    import torch
    n = len(images_dev)
    imgs = [x.contiguous() for x in images_dev]
    if out is None:
        out = torch.empty((n, 3, dst_h, dst_w), dtype=torch.float32, device=imgs[0].device)
    src = (ctypes.c_void_p * n)(*[x.data_ptr() for x in imgs])
    ws = (ctypes.c_int * n)(*[x.shape[1] for x in imgs])
    hs = (ctypes.c_int * n)(*[x.shape[0] for x in imgs])
    check(lib().trtx_letterbox_batch(src, ws, hs, n, _p(out), dst_w, dst_h, _stream()), "trtx_letterbox_batch")
    return out


def preprocess_init(max_image_size, ring_depth=8):
    check(lib().trtx_preprocess_init(max_image_size, ring_depth), "trtx_preprocess_init")


def preprocess_destroy():
    lib().trtx_preprocess_destroy()


def batch_preprocess(images_host, dst_w, dst_h, out):
    """images_host: list of numpy uint8 [H, W, 3]; out: CUDA fp32 [B, 3, dst_h, dst_w]; async on the current stream."""
    n = len(images_host)
    imgs = [np.ascontiguousarray(x, dtype=np.uint8) for x in images_host]
    src = (ctypes.c_void_p * n)(*[x.ctypes.data for x in imgs])
    ws = (ctypes.c_int * n)(*[x.shape[1] for x in imgs])
    hs = (ctypes.c_int * n)(*[x.shape[0] for x in imgs])
    check(lib().trtx_batch_preprocess(src, ws, hs, n, _p(out), dst_w, dst_h, _stream()), "trtx_batch_preprocess")
    return out
