"""ORACLE (test infrastructure): NumPy restatement of the entropy (KL) threshold search TensorRT's IInt8EntropyCalibrator2 is
documented to perform (NVIDIA, "8-bit inference with TensorRT", GTC 2017: 2048-bin histogram of |x|, candidate thresholds at
bins 128..2048, reference distribution clipped into the last bin, candidate = 128-level quantisation spread back over the
non-empty source bins, minimise KL).  TensorRT is closed source and absent here; the reference only feeds it batches
(yolov8/src/calibrator.cpp).  Used to cross-check tensorrtx_amd's calibration."""
import numpy as np


def entropy_threshold(hist, rng, levels=128):
    hist = np.array(hist, dtype=np.float64)
    if len(hist) > 1:
        hist[0] = hist[1]   # NVIDIA pytorch-quantization calib/histogram.py::_compute_amax_entropy ("bins[0] = bins[1]"): exact zeros are
        # representable at any scale; left as a spike in bin 0 they drag the threshold of post-ReLU tensors down to ~1.9 sigma
    nb = len(hist)
    best, best_i = np.inf, nb
    for i in range(levels, nb + 1):
        p = hist[:i].copy()
        p[i - 1] += hist[i:].sum()
        q = np.zeros(i)
        merged = i // levels
        for j in range(levels):
            a = j * merged
            b = i if j == levels - 1 else a + merged
            seg = hist[a:b]
            nz = seg > 0
            if nz.any():
                q[a:b][nz] = seg.sum() / nz.sum()
        sp, sq = p.sum(), q.sum()
        if sp <= 0 or sq <= 0:
            continue
        m = p > 0
        pk = p[m] / sp
        qk = np.where(q[m] > 0, q[m] / sq, 1e-12)
        kl = float((pk * np.log(pk / qk)).sum())
        if kl < best:
            best, best_i = kl, i
    return (best_i + 0.5) * (rng / nb)
