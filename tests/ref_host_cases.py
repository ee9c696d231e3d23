"""Seeded + hand-built decode buffers for the host post-processing pin (reference functions: yolov8/src/postprocess.cpp:71-129,
retinaface/common.hpp:91-130).  Shared by tests/golden/make_ref_golden.py (which runs the REFERENCE'S OWN functions on them,
oracle/_ref/libref_host.so) and tests/test_ref_pinning.py.  Cases avoid full (conf, bbox[0]) ties: the reference sorts with an
unstable std::sort, so their order is unspecified there (the oracle's documented canonicalisation breaks them by slot)."""
import numpy as np

from oracle import det_post as dp
from oracle import yolo_post as yp
from tensorrtx_amd import synth

DET = 90


def _yolo_rows(dets, max_out=1000):
    row = np.zeros((1, 1 + max_out * DET), np.float32)
    row[0, 0] = len(dets)
    for i, d in enumerate(dets):
        row[0, 1 + i * DET:1 + i * DET + 6] = d
    return row


def yolov8_cases():
    cases = {"seeded": yp.decode_c(synth.yolo_head_tensors(3, seed=21), 80, 640, 640, [8, 16, 32])}
    nan = np.float32("nan")
    half = np.float32(0.5)
    above = np.nextafter(half, np.float32(1))
    cases["thresholds_nan_empty"] = np.concatenate([
        _yolo_rows([]),
        _yolo_rows([[0, 0, 10, 10, half, 1], [20, 0, 30, 10, above, 1], [40, 0, 50, 10, nan, 1], [60, 0, 70, 10, 0.9, 2]]),
    ])
    # chain A > B > C on one class (B overlaps A and C, C does not overlap A): greedy keeps A and C
    chain = [[0, 0, 100, 100, 0.9, 3], [40, 0, 140, 100, 0.8, 3], [95, 0, 195, 100, 0.7, 3],
             [40, 0, 140, 100, 0.85, 4],           # same box, other class: untouched
             [300, 300, 300, 300, 0.95, 3], [300, 300, 300, 300, 0.6, 3],   # zero-area pair: 0/0 = NaN is not > thresh
             [500, 0, 600, 100, 0.75, 3], [501, 0, 601, 100, 0.75, 3]]     # conf tie, bbox[0] ascending decides
    cases["chain_ties_degenerate"] = _yolo_rows(chain)
    # IoU exactly at the threshold: two 100x100 boxes offset so that inter/union == 0.45 is not representable -> use
    # boxes whose IoU straddles 0.45 by one representable step
    cases["near_threshold"] = _yolo_rows([[0, 0, 100, 100, 0.9, 0], [0, 37.931034, 100, 137.931034, 0.8, 0],
                                          [200, 0, 300, 100, 0.9, 0], [200, 37.9, 300, 137.9, 0.8, 0],
                                          [400, 0, 500, 100, 0.9, 0], [400, 38.0, 500, 138.0, 0.8, 0]])
    return cases


def retina_cases():
    H, W = 64, 96
    dec = dp.retina_decode(synth.retina_head_tensors(2, H, W, faces=60, seed=31), H, W)
    cases = {"seeded": dec}
    n_f = dec.shape[1]
    hand = np.zeros((2, n_f), np.float32)
    rec = hand[1, 1:].reshape(-1, 15)
    rec[0, :5] = [10, 10, 50, 50, 0.9]
    rec[1, :5] = [12, 12, 52, 52, 0.8]      # suppressed by 0
    rec[2, :5] = [100, 10, 140, 50, 0.100000001]  # == 0.1f: the reference compares with the DOUBLE literal 0.1
    rec[3, :5] = [200, 10, 240, 50, 0.11]
    rec[4, :5] = [300, 300, 300, 300, 0.7]  # zero-area: +1e-6 in the denominator keeps the IoU finite
    rec[5, :5] = [300, 300, 300, 300, 0.6]
    rec[:, 5:] = np.arange(6 * 10, dtype=np.float32).reshape(6, 10) if False else 0
    hand[1, 0] = 6
    cases["hand"] = hand
    return cases


def yolov5_cases():
    """Decode buffers of the anchor-based YoloLayer (38-float records, centre-format boxes) for yolov5/src/postprocess.cpp:30-80."""
    grids = [(80, 80), (40, 40), (20, 20)]
    cases = {"seeded": yp.v5_decode_c(synth.yolov5_head_tensors(3, seed=22), 80, 640, 640, grids, synth.YOLOV5_ANCHORS)}

    def rows(dets, max_out=1000):
        row = np.zeros((1, 1 + max_out * 38), np.float32)
        row[0, 0] = len(dets)
        for i, d in enumerate(dets):
            row[0, 1 + i * 38:1 + i * 38 + 6] = d
        return row
    half = np.float32(0.5)
    cases["thresholds_empty"] = np.concatenate([rows([]), rows([[50, 50, 20, 20, half, 1], [150, 50, 20, 20, np.nextafter(half, np.float32(1)), 1],
                                                                [250, 50, 20, 20, 0.9, 2]])])
    cases["chain_degenerate"] = rows([[50, 50, 100, 100, 0.9, 3], [90, 50, 100, 100, 0.8, 3], [145, 50, 100, 100, 0.7, 3],
                                      [90, 50, 100, 100, 0.85, 4], [300, 300, 0, 0, 0.95, 3], [300, 300, 0, 0, 0.6, 3]])
    return cases
