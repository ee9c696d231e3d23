"""CPU tests of the oracle itself: the C restatement against the independent NumPy / pure-Python
restatements and against hand-computed edge cases (reference: yolov8/plugin/yololayer.cu:178-220,
yolov8/src/postprocess.cpp:71-129)."""
import numpy as np
import pytest

from oracle import yolo_post as yp
from tensorrtx_amd import synth

STRIDES = [8, 16, 32]


def _recs(out, b):
    n = int(out[b, 0])
    return out[b, 1:1 + n * 90].reshape(n, 90)[:, :6]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_decode_c_matches_numpy(seed):
    ins = synth.yolo_head_tensors(3, seed=seed)
    dc = yp.decode_c(ins, 80, 640, 640, STRIDES)
    dn = yp.decode_np(ins, 80, 640, 640, STRIDES)
    assert np.array_equal(dc[:, 0], dn[:, 0])
    for b in range(3):
        a, c = _recs(dc, b), _recs(dn, b)
        assert np.array_equal(a[:, :4], c[:, :4])  # box arithmetic is IEEE basic ops: bit exact
        assert np.array_equal(a[:, 5], c[:, 5])
        assert np.allclose(a[:, 4], c[:, 4], rtol=0, atol=2e-7)  # expf: glibc vs numpy


def test_decode_hand_case():
    # one 2x2 level, stride 8, 2 classes; only cell 3 fires on class 1
    x = np.full((1, 6, 4), -20.0, dtype=np.float32)
    x[0, :4, 3] = [1.0, 2.0, 3.0, 4.0]
    x[0, 5, 3] = 2.0
    out = yp.decode_c([x], 2, 16, 16, [8], max_out=5)
    assert out[0, 0] == 1
    det = out[0, 1:7]
    # row 1, col 1: [(1.5-1)*8, (1.5-2)*8, (1.5+3)*8, (1.5+4)*8]
    assert np.allclose(det[:4], [4.0, -4.0, 36.0, 44.0])
    assert abs(det[4] - 1 / (1 + np.exp(-2.0))) < 1e-6 and det[5] == 1


def test_decode_empty_and_overflow():
    x = np.full((2, 84, 400), -30.0, dtype=np.float32)
    out = yp.decode_c([x], 80, 160, 160, [8])
    assert (out[:, 0] == 0).all()
    x[:, 4, :] = 5.0  # every cell fires -> 400 candidates, cap at 100
    out = yp.decode_c([x], 80, 160, 160, [8], max_out=100)
    assert (out[:, 0] == 100).all()
    # the first 100 cells in canonical order are kept
    rec = out[0, 1:1 + 100 * 90].reshape(100, 90)
    assert np.allclose(rec[:, 0], ((np.arange(100) % 20) + 0.5 - (-30.0)) * 8)


def test_decode_threshold_and_tie():
    # p exactly below/above 0.1; argmax tie -> lowest class index (strict '>')
    x = np.full((1, 7, 4), -30.0, dtype=np.float32)
    x[0, 4, 0] = -2.1972246  # sigmoid ~ 0.1 (just above/below depending on rounding)
    x[0, 4, 1] = -2.3        # < 0.1 -> dropped
    x[0, 5, 2] = 1.0
    x[0, 6, 2] = 1.0         # tie between class 1 and 2 -> class 1
    out = yp.decode_c([x], 3, 16, 16, [8])
    dn = yp.decode_np([x], 3, 16, 16, [8])
    assert out[0, 0] == dn[0, 0]
    n = int(out[0, 0])
    rec = out[0, 1:1 + n * 90].reshape(n, 90)
    assert rec[-1, 5] == 1.0


@pytest.mark.parametrize("seed", [0, 3])
def test_nms_c_matches_python(seed):
    ins = synth.yolo_head_tensors(2, seed=seed)
    dc = yp.decode_c(ins, 80, 640, 640, STRIDES)
    ki, kc, kd = yp.batch_nms_c(dc)
    for b in range(2):
        assert list(ki[b, :kc[b]]) == yp.nms_py(dc[b])
        assert kc[b] > 10


def test_nms_edge_cases():
    out = np.zeros((1, 1 + 1000 * 90), dtype=np.float32)

    def put(i, box, conf, cls):
        out[0, 1 + i * 90:1 + i * 90 + 6] = [*box, conf, cls]

    put(0, [0, 0, 10, 10], 0.9, 1)
    put(1, [1, 1, 11, 11], 0.8, 1)    # iou 0.68 with #0 -> suppressed
    put(2, [1, 1, 11, 11], 0.85, 2)   # other class -> kept
    put(3, [0, 0, 10, 10], 0.5, 1)    # conf == thresh -> dropped (strict >)
    put(4, [50, 50, 60, 60], np.nan, 1)  # NaN dropped
    put(5, [20, 20, 30, 30], 0.7, 1)
    put(6, [19, 20, 30, 30], 0.7, 1)  # same conf: smaller bbox[0] first -> #6 kept, #5 suppressed
    out[0, 0] = 7
    ki, kc, kd = yp.batch_nms_c(out)
    assert list(ki[0, :kc[0]]) == [0, 6, 2]
    assert yp.nms_py(out[0]) == [0, 6, 2]
    # chain A>B>C: B suppressed by A, C overlaps only B -> greedy keeps C
    out[:] = 0
    put(0, [0, 0, 10, 10], 0.9, 0)
    put(1, [4, 0, 14, 10], 0.8, 0)
    put(2, [8, 0, 18, 10], 0.7, 0)
    out[0, 0] = 3
    ki, kc, _ = yp.batch_nms_c(out, nms_thresh=0.4)
    assert list(ki[0, :kc[0]]) == [0, 2]
    out[0, 0] = 0
    ki, kc, _ = yp.batch_nms_c(out)
    assert kc[0] == 0


def test_gpu_mode_postprocess_c_matches_python_and_differs_from_greedy_on_chains():
    """Mode "g" of the reference (postprocess.cu:42-111): C restatement == pure-Python statement; and its non-greedy rule
    really is different from batch_nms on a chain A > B > C where A kills B and B (dead) would still kill C."""
    from tensorrtx_amd import synth
    heads = synth.yolo_head_tensors(2, 80, 320, 320, objects=(20, 60), seed=3)
    dec = yp.decode_c(heads, 80, 320, 320, [8, 16, 32])
    g = yp.gpu_postprocess_c(dec)
    p = np.stack([yp.gpu_postprocess_py(dec[b]) for b in range(2)])
    assert np.array_equal(g, p) and g[:, 0].min() > 50
    # chain: A=(0,0,10,10) .9, B=(4,0,14,10) .8, C=(8,0,18,10) .7, same class: IoU(A,B)=IoU(B,C)=0.43 (>0.4), IoU(A,C)=0.11
    row = np.zeros((1, 1 + 1000 * yp.DET_FLOATS), np.float32)
    row[0, 0] = 3
    for i, (x0, conf) in enumerate(((0, .9), (4, .8), (8, .7))):
        row[0, 1 + i * yp.DET_FLOATS:1 + i * yp.DET_FLOATS + 6] = [x0, 0, x0 + 10, 10, conf, 2]
    keep_flags = yp.gpu_postprocess_c(row, nms_thresh=0.4)[0, 1:].reshape(-1, 7)[:3, 6]
    _, cnt, _ = yp.batch_nms_c(row, nms_thresh=0.4)
    assert keep_flags.tolist() == [1.0, 0.0, 0.0] and cnt[0] == 2  # greedy keeps A and C, mode "g" keeps only A
