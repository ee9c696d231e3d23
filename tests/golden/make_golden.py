#!/usr/bin/env python3
"""Generates the fixtures in this directory.  Runs only where /root/reference is mounted (the build container);
the GPU box and CI use the committed files.

  lenet_ref.wts.gz   weights of the REFERENCE's own PyTorch LeNet (lenet/gen_wts.py:10-45, imported from
                     /root/reference, default nn init under torch.manual_seed(0)) written by the REFERENCE's own .wts
                     writer (the "save to wts" block of lenet/gen_wts.py:83-92, executed from its source, not copied).
  lenet_ref_io.npz   a seeded input batch [2,1,32,32] and the reference model's logits / softmax for it.
  yolo_post_small.npz  decode + NMS results of the C restatement (oracle/csrc/yolo_post_ref.c) on a seeded 160x160
                     head (regression pin for the oracle itself and a fixture for the GPU plugins).

These are the only executable pieces of the reference that can run offline (every other model needs TensorRT, OpenCV,
ultralytics or detectron2): LeNet pins the .wts format, the loader and the builder semantics end to end.
"""
import gzip
import importlib.util
import inspect
import os
import struct
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/lenet/gen_wts.py"


def load_reference_module():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # imported at module scope by the reference, unused here
    spec = importlib.util.spec_from_file_location("ref_lenet_gen_wts", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_module()
    torch.manual_seed(0)
    model = ref.LeNet()
    model.eval()
    x = torch.randn(2, 1, 32, 32, generator=torch.Generator().manual_seed(1))
    with torch.inference_mode():
        logits = model(x)
    # run the reference's own writer block against a temp path
    src = inspect.getsource(ref.main)
    block = src[src.index('with open("../models/lenet.wts"'):]
    tmp = os.path.join(HERE, "lenet_ref.wts")
    block = textwrap.dedent("    " + block.lstrip()).replace('"../models/lenet.wts"', repr(tmp))
    exec(compile(block, REF + ":save-to-wts", "exec"), {"model": model, "struct": struct})
    with open(tmp, "rb") as f, gzip.GzipFile(os.path.join(HERE, "lenet_ref.wts.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    os.remove(tmp)
    np.savez_compressed(os.path.join(HERE, "lenet_ref_io.npz"), x=x.numpy(), logits=logits.numpy(),
                        prob=torch.softmax(logits, 1).numpy())

    # oracle regression pin: YOLO decode + NMS on a seeded small head
    from oracle import yolo_post as yp
    from tensorrtx_amd import synth
    heads = synth.yolo_head_tensors(2, 80, 160, 160, objects=(10, 30), seed=7)
    dec = yp.decode_c(heads, 80, 160, 160, [8, 16, 32])
    keep_idx, keep_cnt, keep_det = yp.batch_nms_c(dec)
    np.savez_compressed(os.path.join(HERE, "yolo_post_small.npz"), counts=dec[:, 0].copy(), keep_idx=keep_idx,
                        keep_cnt=keep_cnt, keep_det=keep_det[:, :64].copy())
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
