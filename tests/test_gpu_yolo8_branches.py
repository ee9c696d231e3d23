"""f4: the seg / pose / obb branches of the YOLOv8 YoloLayer (yolov8/plugin/yololayer.cu:222-279) and the oriented-box
post-processing (host nms_obb with ProbIoU, yolov8/src/postprocess.cpp:303-393; GPU mode decode_kernel_obb / nms_kernel_obb,
yolov8/src/postprocess.cu:7-40,113-166): product HIP kernels vs the C oracle vs the reference's own code."""
import ctypes
import struct

import numpy as np
import pytest

from oracle import ref
from oracle import yolo_post as yp
from tensorrtx_amd import capi, synth

STRIDES = [8, 16, 32]


def branch_inputs(batch, seed, classes=80, size=640, nk=17, seg=False, pose=False, obb=False):
    base = synth.yolo_head_tensors(batch, classes=classes, net_h=size, net_w=size, seed=seed)
    rng = np.random.default_rng(1000 + seed)
    outs = []
    for x in base:
        extra = []
        cells = x.shape[2]
        if seg:
            extra.append(rng.normal(0, 1, size=(batch, 32, cells)))
        if pose:
            k = rng.normal(0, 1.5, size=(batch, nk * 3, cells))
            k[:, 2::3] = rng.normal(0.5, 2.0, size=(batch, nk, cells))
            extra.append(k)
        if obb:
            extra.append(rng.normal(0, 2.0, size=(batch, 1, cells)))
        # channel order of the input tensor: 4 box, classes, then (obb angle | pose | seg) per the index arithmetic of CalDetection:
        # seg coefficients come last, pose triples before them, the obb angle right after the classes
        order = []
        if obb:
            order.append(extra[-1])
        if pose:
            order.append(extra[1 if seg else 0])
        if seg:
            order.append(extra[0])
        outs.append(np.concatenate([x] + order, axis=1).astype(np.float32))
    return outs


def _cmp(got, ref_, obb=False):
    assert np.array_equal(got[:, 0], ref_[:, 0])
    for b in range(ref_.shape[0]):
        n = int(ref_[b, 0])
        g = got[b, 1:1 + n * 90].reshape(n, 90)
        r = ref_[b, 1:1 + n * 90].reshape(n, 90)
        assert np.array_equal(g[:, 5], r[:, 5])
        assert np.array_equal(g[:, 6:38], r[:, 6:38]), "mask coefficients are copied"
        if obb:  # cos/sin of a double angle: device vs host libm, then one rounding to float
            assert np.allclose(g[:, :4], r[:, :4], rtol=1e-6, atol=1e-4) and np.allclose(g[:, 89], r[:, 89], rtol=1e-6, atol=1e-7)
        else:
            assert np.array_equal(g[:, :4], r[:, :4])
        assert np.allclose(g[:, 4], r[:, 4], rtol=0, atol=2e-7)
        gk, rk = g[:, 38:89], r[:, 38:89]
        assert np.array_equal(gk == -1, rk == -1) and np.allclose(gk, rk, rtol=1e-6, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [dict(seg=True), dict(pose=True), dict(obb=True), dict(seg=True, pose=True, obb=True)])
@pytest.mark.parametrize("kpt_conf", [0.0, 0.5])
def test_decode_branches_match_oracle(gpu, flags, kpt_conf):
    import torch
    ins = branch_inputs(3, seed=3, **flags)
    got = capi.yolo_decode_ex([torch.from_numpy(x).to(gpu) for x in ins], 80, 640, 640, STRIDES, 1000, 17, kpt_conf, **flags).cpu().numpy()
    want = yp.decode_ex_c(ins, 80, 640, 640, STRIDES, 1000, 17, kpt_conf, **flags)
    _cmp(got, want, obb=flags.get("obb", False))
    if flags.get("pose"):
        k = want[0, 1:1 + int(want[0, 0]) * 90].reshape(-1, 90)[:, 38:89]
        assert (k == -1).any() and (k != -1).any()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [dict(seg=True), dict(pose=True), dict(obb=True)])
def test_decode_branches_equal_the_reference_plugin(gpu, flags):
    """the unmodified reference YoloLayerPlugin (hipcc build) created with isSeg / isPose / isObb vs the built-in plugin"""
    import torch
    import ref_cases as rc
    if not ref.available("libref_yolov8_plugin.so"):
        pytest.skip("oracle/_ref/libref_yolov8_plugin.so not present")
    ins = branch_inputs(2, seed=5, **flags)
    info = np.array([80, 17, 0, 640, 640, 1000, int(flags.get("seg", 0)), int(flags.get("pose", 0)), int(flags.get("obb", 0)), 8, 16, 32], np.int32)
    case = rc.Case("yolov8_branch", "yolov8_plugin", "YoloLayer_TRT", 2, ins, [(2, 1 + 1000 * 90)], fields=[("combinedInfo", info)])
    want = rc.run_reference(case, gpu)[0]
    builtin = ref.registry_get("YoloLayer_TRT")
    v = ref.make_plugin(builtin, fields=[("combinedInfo", info)])
    got = ref.run_plugin(v, 2, [torch.from_numpy(x).to(gpu) for x in ins], [(2, 1 + 1000 * 90)])[0].cpu().numpy()
    a, b = rc.canon_records(got, 90, 90), rc.canon_records(want, 90, 90)
    for ra, rb in zip(a, b):
        assert ra.shape == rb.shape and ra.shape[0] > 100
        assert np.allclose(ra, rb, rtol=1e-6, atol=1e-4)
        if not flags.get("obb"):
            assert np.array_equal(ra[:, :4], rb[:, :4])


def _obb_decode(batch, seed):
    return yp.decode_ex_c(branch_inputs(batch, seed, classes=15, obb=True), 15, 640, 640, STRIDES, 1000, 17, 0.0, obb=True)


def test_oracle_nms_obb_equals_the_reference_host_function():
    if not ref.available("libref_host.so"):
        pytest.skip("oracle/_ref/libref_host.so not present")
    dec = _obb_decode(3, seed=9)
    ki, kc, kd = yp.batch_nms_obb_c(dec)
    for b in range(3):
        want = ref.yolov8_nms_obb(dec[b])
        assert kc[b] == len(want) and kc[b] > 20
        assert np.array_equal(kd[b, :kc[b], :6], want[:, :6]) and np.array_equal(kd[b, :kc[b], 6], want[:, 89])


@pytest.mark.gpu
def test_nms_obb_matches_oracle_and_reference(gpu):
    import torch
    dec = _obb_decode(4, seed=10)
    ri, rc_, rd = yp.batch_nms_obb_c(dec)
    gi, gc, gd = capi.yolo_nms_obb(torch.from_numpy(dec).to(gpu))
    gi, gc, gd = gi.cpu().numpy(), gc.cpu().numpy(), gd.cpu().numpy()
    assert np.array_equal(gc, rc_)
    for b in range(4):
        assert np.array_equal(gi[b, :rc_[b]], ri[b, :rc_[b]]) and np.array_equal(gd[b, :rc_[b]], rd[b, :rc_[b]])
        if ref.available("libref_host.so"):
            want = ref.yolov8_nms_obb(dec[b])
            assert np.array_equal(gd[b, :gc[b], :6], want[:, :6])


@pytest.mark.gpu
def test_gpu_mode_obb_equals_oracle_and_reference_kernels(gpu):
    import torch
    dec = _obb_decode(3, seed=11)
    d = torch.from_numpy(dec).to(gpu)
    got = capi.yolo_postprocess_gpu_obb(d).cpu().numpy()
    want = yp.gpu_postprocess_obb_c(dec)
    assert np.array_equal(got[:, 0], want[:, 0])
    n = int(want[:, 0].max())
    g, w = got[:, 1:1 + n * 8].reshape(3, n, 8), want[:, 1:1 + n * 8].reshape(3, n, 8)
    assert np.array_equal(g[..., :6], w[..., :6]) and np.array_equal(g[..., 7], w[..., 7])
    assert (g[..., 6] == w[..., 6]).mean() > 0.995   # keep flags: device vs host powf/logf may flip a pair sitting on the threshold
    # NOT compared with the reference's own cuda_decode_obb / cuda_nms_obb: decode_kernel_obb writes 8 floats per record at a
    # stride of bbox_element = 7 (types.h:18-19, postprocess.cu:31-39), so each record's angle lands on the
    # next record's cx in atomicAdd order and nms_kernel_obb reads pcurrent[7] from there: its output depends on thread timing.
    # The product (and the oracle) implement the evident intent with 8-float records.
    if ref.available("libref_yolov8_post.so"):
        L = ref.family_lib("yolov8_post")
        out = torch.zeros(1 + 1000 * 8, dtype=torch.float32, device=gpu)
        L.ref_yolov8_gpu_postprocess_obb(ctypes.c_void_p(d[0].data_ptr()), 1000, ctypes.c_float(0.5), ctypes.c_float(0.45),
                                         ctypes.c_void_p(out.data_ptr()), 1000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert out[0].item() == got[0, 0]  # the candidate counter is the one thing the overlap cannot corrupt


def test_builtin_yololayer_blob_round_trip_with_branches():
    """CPU: blob layout int classCount, nKpt, float kptConf, int threadCount, netW, netH, maxOut, nStrides, strides[], bool seg, pose, obb
    (yololayer.cu:75-101)."""
    c = ref.registry_get("YoloLayer_TRT")
    blob = struct.pack("<iifiiiii", 15, 17, 0.5, 256, 640, 640, 1000, 3) + struct.pack("<iii", 8, 16, 32) + struct.pack("<???", False, True, True)
    v = ref.make_plugin(c, blob=blob)
    assert ref.plugin_blob(v) == blob
    assert v.initialize(v.self) == 0
    v.destroy(v.self)
