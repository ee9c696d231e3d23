"""ORACLE — test infrastructure only.

CPU restatements of the wang-xinyu/tensorrtx algorithms on the hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product (``tensorrtx_amd``) never imports it.

Parity status (round 2): PINNED on the reference's own code for every plugin / post-processing row.
``oracle/ref_build.py`` compiles, from the sources where they lie under /root/reference (nothing copied into git),
  * the reference's host NMS functions (yolov8 postprocess.cpp, yolov5 postprocess.cpp, retinaface common.hpp) with g++,
  * the reference's CUDA plugins (yolov8/yolov5 YoloLayer, RetinaFace Decode, the six R-CNN plugins) and
    yolov8 postprocess.cu / preprocess.cu, unmodified, with hipcc as *user plugins* against include/NvInfer.h,
into ``oracle/_ref/`` (git-ignored, travels to the GPU box).  tests/test_ref_pinning.py checks the C restatements in
``oracle/csrc`` against them on seeded + edge cases and against their committed outputs (tests/golden/ref_host_nms.npz,
ref_plugins.npz: produced by the reference kernels running on the MI355X).  The `.wts` reader and LeNet are pinned on
fixtures generated from the reference's runnable Python (tests/golden/make_golden.py).  Still unpinned: the conv
arithmetic of the YOLOv8 / ResNet / RetinaFace / R-CNN *graphs* (it lives in closed TensorRT; weights and datasets are
absent) — those are anchored on the reference builder source only.  Each function cites the file:line it restates.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile the plain-C restatements (gcc) into oracle/_build/liboracle.so."""
    so = os.path.join(_HERE, "_build", "liboracle.so")
    srcs = [os.path.join(_HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(_HERE, "csrc")))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "_build/liboracle.so"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB
