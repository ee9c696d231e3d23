"""ORACLE (test infrastructure): the reference's GPU letterbox pre-processing (yolov8/src/preprocess.cu:7-117) restated in C
(oracle/csrc/letterbox_ref.c).  Pinned on the reference's own kernel in tests/test_gpu_letterbox.py."""
import ctypes

import numpy as np

from . import lib


def letterbox_matrix(src_w, src_h, dst_w, dst_h):
    m = np.zeros(6, np.float32)
    lib().letterbox_matrix_ref(src_w, src_h, dst_w, dst_h, m.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return m


def letterbox(img_hwc_bgr_u8, dst_w, dst_h):
    """uint8 [H, W, 3] BGR -> fp32 [3, dst_h, dst_w] RGB / 255, border 128."""
    img = np.ascontiguousarray(img_hwc_bgr_u8, dtype=np.uint8)
    h, w = img.shape[:2]
    out = np.zeros((3, dst_h, dst_w), np.float32)
    lib().letterbox_ref(img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), w, h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), dst_w, dst_h)
    return out
