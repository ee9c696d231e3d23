"""GPU parity: HIP YoloLayer decode + NMS (through the C ABI) against the oracle.
Reference: yolov8/plugin/yololayer.cu:178-316, yolov8/src/postprocess.cpp:71-129."""
import numpy as np
import pytest

from oracle import yolo_post as yp
from tensorrtx_amd import capi, synth

pytestmark = pytest.mark.gpu
STRIDES = [8, 16, 32]


def _decode_gpu(ins, dev, classes=80, h=640, w=640, strides=STRIDES, max_out=1000):
    import torch
    t = [torch.from_numpy(x).to(dev) for x in ins]
    out = capi.yolo_decode(t, classes, h, w, strides, max_out)
    torch.cuda.synchronize()
    return out


def _compare_decode(got, ref, max_out=1000):
    assert np.array_equal(got[:, 0], ref[:, 0]), (got[:, 0], ref[:, 0])
    for b in range(ref.shape[0]):
        n = int(ref[b, 0])
        g = got[b, 1:1 + n * 90].reshape(n, 90)[:, :6]
        r = ref[b, 1:1 + n * 90].reshape(n, 90)[:, :6]
        assert np.array_equal(g[:, :4], r[:, :4]), "bbox must be bit-exact (IEEE basic ops)"
        assert np.array_equal(g[:, 5], r[:, 5]), "class ids"
        # conf = 1/(1+expf(-x)): device expf vs glibc expf may differ by an ulp
        assert np.allclose(g[:, 4], r[:, 4], rtol=0, atol=2e-7)


@pytest.mark.parametrize("batch,seed", [(1, 0), (4, 1), (32, 2)])
def test_decode_matches_oracle(gpu, batch, seed):
    ins = synth.yolo_head_tensors(batch, seed=seed)
    got = _decode_gpu(ins, gpu).cpu().numpy()
    ref = yp.decode_c(ins, 80, 640, 640, STRIDES)
    _compare_decode(got, ref)


def test_decode_overflow_and_empty(gpu):
    x = np.full((2, 84, 400), -30.0, dtype=np.float32)
    got = _decode_gpu([x], gpu, h=160, w=160, strides=[8]).cpu().numpy()
    assert (got[:, 0] == 0).all()
    x[:, 4, :] = 5.0
    x[:, :4, :] = np.random.default_rng(0).uniform(0, 5, size=(2, 4, 400)).astype(np.float32)
    got = _decode_gpu([x], gpu, h=160, w=160, strides=[8], max_out=100).cpu().numpy()
    ref = yp.decode_c([x], 80, 160, 160, [8], max_out=100)
    _compare_decode(got, ref, 100)


def test_decode_ragged_levels_scalar_path(gpu):
    # 5x7 and 3x3 grids: cell counts not multiples of 4 -> scalar kernel, odd class count
    rng = np.random.default_rng(5)
    a = rng.normal(-2, 2, size=(3, 4 + 7, 35)).astype(np.float32)
    b = rng.normal(-2, 2, size=(3, 4 + 7, 9)).astype(np.float32)
    # net 40x56 with stride 8 -> 5x7 ; stride 16 -> 2x3 (not 3x3) so use matching shapes
    b = b[:, :, :6]
    got = _decode_gpu([a, b], gpu, classes=7, h=40, w=56, strides=[8, 16], max_out=50).cpu().numpy()
    ref = yp.decode_c([a, b], 7, 40, 56, [8, 16], max_out=50)
    _compare_decode(got, ref, 50)


@pytest.mark.parametrize("batch,seed", [(1, 0), (8, 1), (32, 2)])
def test_nms_bit_exact_on_identical_inputs(gpu, batch, seed):
    import torch
    ins = synth.yolo_head_tensors(batch, seed=seed)
    dec = yp.decode_c(ins, 80, 640, 640, STRIDES)  # identical input for both sides
    ki, kc, kd = yp.batch_nms_c(dec)
    gi, gc, gd = capi.yolo_nms(torch.from_numpy(dec).to(gpu))
    torch.cuda.synchronize()
    gi, gc, gd = gi.cpu().numpy(), gc.cpu().numpy(), gd.cpu().numpy()
    assert np.array_equal(gc, kc)
    for b in range(batch):
        assert np.array_equal(gi[b, :kc[b]], ki[b, :kc[b]])
        assert np.array_equal(gd[b, :kc[b]], kd[b, :kc[b]])
    assert kc.min() > 5


def test_nms_edge_cases(gpu):
    import torch
    out = np.zeros((3, 1 + 1000 * 90), dtype=np.float32)

    def put(b, i, box, conf, cls):
        out[b, 1 + i * 90:1 + i * 90 + 6] = [*box, conf, cls]

    put(0, 0, [0, 0, 10, 10], 0.9, 1)
    put(0, 1, [1, 1, 11, 11], 0.8, 1)
    put(0, 2, [1, 1, 11, 11], 0.85, 2)
    put(0, 3, [0, 0, 10, 10], 0.5, 1)
    put(0, 4, [50, 50, 60, 60], np.nan, 1)
    put(0, 5, [20, 20, 30, 30], 0.7, 1)
    put(0, 6, [19, 20, 30, 30], 0.7, 1)
    out[0, 0] = 7
    put(1, 0, [0, 0, 10, 10], 0.9, 0)
    put(1, 1, [4, 0, 14, 10], 0.8, 0)
    put(1, 2, [8, 0, 18, 10], 0.7, 0)
    out[1, 0] = 3
    out[2, 0] = 0
    gi, gc, _ = capi.yolo_nms(torch.from_numpy(out).to(gpu), nms_thresh=0.45)
    gi, gc = gi.cpu().numpy(), gc.cpu().numpy()
    assert list(gi[0, :gc[0]]) == [0, 6, 2]
    assert gc[2] == 0
    ki, kc, _ = yp.batch_nms_c(out, nms_thresh=0.45)
    assert np.array_equal(gc, kc)
    assert list(gi[1, :gc[1]]) == list(ki[1, :kc[1]])


def test_nms_full_buffer_dense_clusters(gpu):
    """1000 candidates, few classes, heavy overlap: exercises every 64-box block of the blocked greedy pass."""
    import torch
    rng = np.random.default_rng(11)
    B = 4
    out = np.zeros((B, 1 + 1000 * 90), dtype=np.float32)
    for b in range(B):
        n = 1000
        cx, cy = rng.uniform(50, 590, n), rng.uniform(50, 590, n)
        w, h = rng.uniform(20, 120, n), rng.uniform(20, 120, n)
        rec = out[b, 1:].reshape(1000, 90)
        rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
        rec[:, 4] = np.round(rng.uniform(0.3, 1.0, n), 2)  # many exact conf ties
        rec[:, 5] = rng.integers(0, 3, n)
        out[b, 0] = n
    ki, kc, kd = yp.batch_nms_c(out)
    gi, gc, gd = capi.yolo_nms(torch.from_numpy(out).to(gpu))
    gi, gc = gi.cpu().numpy(), gc.cpu().numpy()
    assert np.array_equal(gc, kc)
    for b in range(B):
        assert np.array_equal(gi[b, :kc[b]], ki[b, :kc[b]])


def test_decode_then_nms_end_to_end_properties(gpu):
    """Full-size (batch 32) properties: idempotence of NMS and sortedness of the emission order."""
    import torch
    ins = synth.yolo_head_tensors(32, seed=7)
    dec = _decode_gpu(ins, gpu)
    gi, gc, gd = capi.yolo_nms(dec)
    torch.cuda.synchronize()
    gc_h, gd_h = gc.cpu().numpy(), gd.cpu().numpy()
    # re-running NMS on the survivors keeps all of them (idempotence)
    buf = np.zeros((32, 1 + 1000 * 90), dtype=np.float32)
    for b in range(32):
        n = gc_h[b]
        buf[b, 0] = n
        buf[b, 1:].reshape(1000, 90)[:n, :6] = gd_h[b, :n]
        cls, conf = gd_h[b, :n, 5], gd_h[b, :n, 4]
        order = np.lexsort((-conf, cls))
        assert np.array_equal(order, np.arange(n)) or np.all(np.diff(cls) >= 0)
        assert (conf > 0.5).all()
    _, gc2, _ = capi.yolo_nms(torch.from_numpy(buf).to(gpu))
    assert np.array_equal(gc2.cpu().numpy(), gc_h)


def test_gpu_mode_postprocess_matches_oracle(gpu):
    """trtx_yolo_postprocess_gpu == the sequential restatement of the reference's mode "g" (bit exact, input slot order)."""
    import torch
    heads = synth.yolo_head_tensors(3, 80, 640, 640, objects=(60, 200), seed=21)
    dec = yp.decode_c(heads, 80, 640, 640, [8, 16, 32])
    dec[2, 0] = 0  # an empty image
    ref = yp.gpu_postprocess_c(dec)
    got = capi.yolo_postprocess_gpu(torch.from_numpy(dec).to(gpu)).cpu().numpy()
    assert np.array_equal(got, ref)
    assert ref[0, 1:].reshape(-1, 7)[:, 6].sum() > 20
