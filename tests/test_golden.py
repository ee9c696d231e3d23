"""Fixtures generated from the reference's own runnable Python (tests/golden/make_golden.py): the PyTorch LeNet of
lenet/gen_wts.py with weights written by the reference's own .wts writer, and its outputs on a seeded batch.
They pin the .wts format, both loaders (oracle parser, C++ trtx_wts_load), the oracle's LeNet restatement and the C++
host builder against something the reference itself computed.  `yolo_post_small.npz` pins the C post-processing oracle."""
import ctypes
import gzip
import os

import numpy as np
import pytest
import torch

from oracle import graph_interp as gi
from oracle import models_torch as mt
from oracle import wts as owts
from oracle import yolo_post as yp
from tensorrtx_amd import capi, engine, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def lenet_wts(tmp_path_factory):
    p = tmp_path_factory.mktemp("golden") / "lenet_ref.wts"
    with gzip.open(os.path.join(GOLD, "lenet_ref.wts.gz"), "rb") as g:
        p.write_bytes(g.read())
    return str(p)


def test_oracle_parser_and_lenet_restatement_reproduce_the_reference_outputs(lenet_wts):
    io = np.load(os.path.join(GOLD, "lenet_ref_io.npz"))
    tensors = owts.load_wts(lenet_wts)
    assert list(tensors) == ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "fc1.weight", "fc1.bias",
                             "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias"]
    with torch.inference_mode():
        prob = mt.lenet(mt.Params(tensors), torch.from_numpy(io["x"])).numpy()
    assert np.allclose(prob, io["prob"], atol=1e-6)
    assert np.array_equal(prob.argmax(1), io["logits"].argmax(1))


def test_cpp_loader_reads_the_reference_written_file_bit_exactly(lenet_wts):
    ref = owts.load_wts(lenet_wts)
    L = capi.lib()
    w = ctypes.c_void_p()
    assert L.trtx_wts_load(lenet_wts.encode(), ctypes.byref(w)) == 0
    assert L.trtx_wts_count(w) == len(ref)
    for i, (name, arr) in enumerate(ref.items()):
        n, v, c = ctypes.c_char_p(), ctypes.POINTER(ctypes.c_float)(), ctypes.c_int64()
        assert L.trtx_wts_entry(w, i, ctypes.byref(n), ctypes.byref(v), ctypes.byref(c)) == 0
        assert n.value.decode() == name and c.value == arr.size
        assert np.ctypeslib.as_array(v, shape=(c.value,)).tobytes() == arr.tobytes()
    L.trtx_wts_free(w)


def test_cpp_lenet_builder_on_reference_weights_matches_reference_outputs(lenet_wts):
    io = np.load(os.path.join(GOLD, "lenet_ref_io.npz"))
    # N = 1 like the reference: its MatMul(W, x^T) -> reshape(-1, C) chain (lenet.cpp:88-121) only means "per-sample FC"
    # for a single sample, so the two golden inputs are run one at a time
    plan = engine.build_plan("lenet", lenet_wts, batch=1)
    desc = engine.describe_plan(plan)
    for i in range(2):
        out = gi.run(desc, plan, {"data": io["x"][i:i + 1]})["prob"].reshape(10).numpy()
        assert np.allclose(out, io["prob"][i], atol=1e-6)


def test_post_processing_oracle_is_stable():
    g = np.load(os.path.join(GOLD, "yolo_post_small.npz"))
    heads = synth.yolo_head_tensors(2, 80, 160, 160, objects=(10, 30), seed=7)
    dec = yp.decode_c(heads, 80, 160, 160, [8, 16, 32])
    keep_idx, keep_cnt, keep_det = yp.batch_nms_c(dec)
    assert np.array_equal(dec[:, 0], g["counts"]) and np.array_equal(keep_cnt, g["keep_cnt"])
    assert np.array_equal(keep_idx, g["keep_idx"]) and np.array_equal(keep_det[:, :64], g["keep_det"])


@pytest.mark.gpu
def test_gpu_lenet_engine_on_reference_weights(gpu, lenet_wts):
    from test_gpu_engine import _run
    io = np.load(os.path.join(GOLD, "lenet_ref_io.npz"))
    plan = engine.build_plan("lenet", lenet_wts, batch=1)
    for i in range(2):
        out = _run(plan, {"data": io["x"][i:i + 1]}, 1, gpu)["prob"].reshape(10).numpy()
        assert np.abs(out - io["prob"][i]).max() < 1e-5


@pytest.mark.gpu
def test_gpu_yolo_plugins_reproduce_the_fixture(gpu):
    g = np.load(os.path.join(GOLD, "yolo_post_small.npz"))
    heads = synth.yolo_head_tensors(2, 80, 160, 160, objects=(10, 30), seed=7)
    dec = capi.yolo_decode([torch.from_numpy(h).to(gpu) for h in heads], 80, 160, 160, [8, 16, 32])
    ki, kc, kd = capi.yolo_nms(dec)
    assert np.array_equal(dec[:, 0].cpu().numpy(), g["counts"]) and np.array_equal(kc.cpu().numpy(), g["keep_cnt"])
    kept = [set(map(tuple, np.round(kd[b, :int(kc[b])].cpu().numpy(), 4))) for b in range(2)]
    want = [set(map(tuple, np.round(g["keep_det"][b, :int(g["keep_cnt"][b])], 4))) for b in range(2)]
    assert kept == want  # decode slots differ (atomics vs scan order), the kept detections do not
