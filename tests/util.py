"""Shared helpers for the tests: seeded synthetic .wts files (cached under /tmp) and plan building."""
import functools
import hashlib
import os

import numpy as np
import torch

from oracle import models_torch as mt
from tensorrtx_amd import wts as wts_writer

CACHE = os.environ.get("TRTX_TEST_CACHE", "/tmp/trtx_test_cache")
WTS_VERSION = {"retinaface_r50": 2, "rcnn_r50c4": 3}  # bump when a model's synthetic initialisation changes (cache key)


def synth_wts(model: str, seed: int = 0, dialect: str = "double", **kw):
    """Create (once) the seeded synthetic weights of `model`; returns (path, OrderedDict of tensors)."""
    os.makedirs(CACHE, exist_ok=True)
    tag = hashlib.sha1(repr((model, seed, dialect, sorted(kw.items()), WTS_VERSION.get(model, 1))).encode()).hexdigest()[:12]
    path = os.path.join(CACHE, f"{model}_{tag}.wts")
    fn, x = {
        "lenet": (mt.lenet, torch.zeros(1, 1, 32, 32)),
        "resnet50": (mt.resnet50, torch.zeros(1, 3, 64, 64)),
        "yolov8n": (mt.yolov8_det, torch.zeros(1, 3, 64, 64)),
        "yolov8n_seg": (functools.partial(mt.yolov8_det, task="seg"), torch.zeros(1, 3, 64, 64)),
        "yolov8n_pose": (functools.partial(mt.yolov8_det, task="pose", num_class=1), torch.zeros(1, 3, 64, 64)),
        "yolov8n_obb": (functools.partial(mt.yolov8_det, task="obb", num_class=15), torch.zeros(1, 3, 64, 64)),
        "retinaface_r50": (mt.retinaface_r50, torch.zeros(1, 3, 64, 64)),
        "rcnn_r50c4": (functools.partial(mt.rcnn_r50c4, stage="init"), torch.zeros(1, 64, 64, 3)),
    }[model]
    tensors, _ = mt.make_weights(fn, x, seed=seed, **kw)
    if not os.path.exists(path):
        tmp = f"{path}.{os.getpid()}.tmp"  # ranks of a multi-GPU bench may generate the same file concurrently
        wts_writer.write_wts(tmp, {k: v.numpy() for k, v in tensors.items()}, dialect=dialect)
        os.replace(tmp, path)
    return path, tensors
