"""f1: anchor-based YoloLayer (YOLOv5 family) decode + centre-format NMS on the GPU, through the C ABI, against the oracle
(oracle/csrc/yolov5_post_ref.c; itself pinned on the reference's plugin and host nms in test_ref_pinning.py).
Reference: yolov5/plugin/yololayer.cu:161-227, yolov5/src/postprocess.cpp:30-80."""
import ctypes
import struct

import numpy as np
import pytest

from oracle import yolo_post as yp
from tensorrtx_amd import capi, synth

GRIDS = [(80, 80), (40, 40), (20, 20)]


def _cmp_decode(got, ref, seg=False):
    assert np.array_equal(got[:, 0], ref[:, 0]), (got[:, 0], ref[:, 0])
    for b in range(ref.shape[0]):
        n = int(ref[b, 0])
        g = got[b, 1:1 + n * 38].reshape(n, 38)
        r = ref[b, 1:1 + n * 38].reshape(n, 38)
        assert np.array_equal(g[:, 5], r[:, 5]), "class ids"
        assert np.allclose(g[:, :5], r[:, :5], rtol=2e-6, atol=2e-6)  # every value passes through expf (device vs glibc: 1 ulp)
        if seg:
            assert np.array_equal(g[:, 6:], r[:, 6:]), "mask coefficients are copied"


@pytest.mark.gpu
@pytest.mark.parametrize("batch,seed,seg", [(1, 0, False), (4, 1, False), (32, 2, False), (3, 3, True)])
def test_v5_decode_matches_oracle(gpu, batch, seed, seg):
    import torch
    ins = synth.yolov5_head_tensors(batch, seed=seed, seg=seg)
    got = capi.yolov5_decode([torch.from_numpy(x).to(gpu) for x in ins], 80, 640, 640, GRIDS, synth.YOLOV5_ANCHORS, 1000, seg).cpu().numpy()
    ref = yp.v5_decode_c(ins, 80, 640, 640, GRIDS, synth.YOLOV5_ANCHORS, 1000, seg)
    _cmp_decode(got, ref, seg)
    assert ref[:, 0].min() > 50


@pytest.mark.gpu
def test_v5_decode_overflow_empty_and_ragged_grid(gpu):
    import torch
    grids = [(7, 5), (3, 2)]   # ragged, non-square grids; 4 classes
    anchors = [[4, 5, 8, 9, 12, 7], [20, 30, 25, 18, 40, 44]]
    rng = np.random.default_rng(4)
    ins = [rng.normal(0, 2, size=(3, 3 * 9, gw * gh)).astype(np.float32) for gw, gh in grids]
    ins[0][1] = -30.0  # image 1: nothing passes on level 0
    ins[1][1] = -30.0
    ins[0][2, 4::9] = 8.0  # image 2: every anchor of every cell passes -> overflows max_out = 40
    got = capi.yolov5_decode([torch.from_numpy(x).to(gpu) for x in ins], 4, 40, 56, grids, anchors, 40).cpu().numpy()
    ref = yp.v5_decode_c(ins, 4, 40, 56, grids, anchors, 40)
    assert ref[1, 0] == 0 and ref[2, 0] == 40
    assert np.array_equal(got[:, 0], ref[:, 0])
    for b in range(3):
        n = int(ref[b, 0])
        assert np.allclose(got[b, 1:1 + n * 38], ref[b, 1:1 + n * 38], rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,seed", [(1, 5), (32, 6)])
def test_v5_nms_bit_exact(gpu, batch, seed):
    import torch
    dec = yp.v5_decode_c(synth.yolov5_head_tensors(batch, seed=seed), 80, 640, 640, GRIDS, synth.YOLOV5_ANCHORS)
    ri, rc, rd = yp.v5_batch_nms_c(dec)
    gi, gc, gd = capi.yolov5_nms(torch.from_numpy(dec).to(gpu))
    gi, gc, gd = gi.cpu().numpy(), gc.cpu().numpy(), gd.cpu().numpy()
    assert np.array_equal(gc, rc) and rc.min() > 20
    for b in range(batch):
        assert np.array_equal(gi[b, :rc[b]], ri[b, :rc[b]])
        assert np.array_equal(gd[b, :rc[b]], rd[b, :rc[b]])


@pytest.mark.gpu
def test_v5_nms_ties_thresholds_and_chains(gpu):
    import torch
    import ref_host_cases as hc
    for name, rows in hc.yolov5_cases().items():
        ri, rc, rd = yp.v5_batch_nms_c(rows)
        gi, gc, gd = capi.yolov5_nms(torch.from_numpy(rows).to(gpu))
        assert np.array_equal(gc.cpu().numpy(), rc), name
        for b in range(rows.shape[0]):
            assert np.array_equal(gi.cpu().numpy()[b, :rc[b]], ri[b, :rc[b]]), name
    # full (class, conf) ties: slot order decides (documented canonicalisation)
    row = np.zeros((1, 1 + 1000 * 38), np.float32)
    row[0, 0] = 50
    rec = row[0, 1:].reshape(-1, 38)
    rec[:50, :4] = [100, 100, 40, 40]
    rec[:50, 4] = 0.8
    rec[25:50, 0] += 200
    ri, rc, _ = yp.v5_batch_nms_c(row)
    gi, gc, _ = capi.yolov5_nms(torch.from_numpy(row).to(gpu))
    assert rc[0] == 2 and list(ri[0, :2]) == [0, 25]
    assert np.array_equal(gi.cpu().numpy()[0, :2], ri[0, :2])


def test_builtin_yololayer_accepts_the_anchor_based_parameter_set():
    """CPU: the built-in "YoloLayer_TRT" creator builds the anchor-based plugin from the yolov5 PluginFields
    ("netinfo" + "kernels", yolov5/src/model.cpp:246-275) and round-trips the reference's blob layout (yololayer.cu:49-89)."""
    from oracle import ref
    import ref_cases as rc
    c = ref.registry_get("YoloLayer_TRT")
    kern = np.zeros(3, dtype=rc.YOLO5_KERNEL)
    for i, (gw, gh) in enumerate(GRIDS):
        kern[i] = (gw, gh, synth.YOLOV5_ANCHORS[i])
    v = ref.make_plugin(c, fields=[("netinfo", np.array([80, 640, 640, 1000, 0], np.int32)), ("kernels", kern.view(np.uint8), 3)])
    blob = ref.plugin_blob(v)
    want = struct.pack("<iiiiii?", 80, 256, 3, 640, 640, 1000, False) + kern.tobytes()
    assert blob == want
    d = ref.Dims()
    assert v.get_output_dims(v.self, 0, None, 0, ctypes.byref(d)) == 0 and d.d[0] == 1000 * 38 + 1
    v2 = ref.make_plugin(c, blob=blob)
    assert ref.plugin_blob(v2) == blob
    v.destroy(v.self)
    v2.destroy(v2.self)
    # the YOLOv8 parameter set still resolves to the anchor-free plugin
    v8 = ref.make_plugin(c, fields=[("combinedInfo", np.array([80, 17, 0, 640, 640, 1000, 0, 0, 0, 8, 16, 32], np.int32))])
    assert v8.get_output_dims(v8.self, 0, None, 0, ctypes.byref(d)) == 0 and d.d[0] == 1000 * 90 + 1
    v8.destroy(v8.self)
