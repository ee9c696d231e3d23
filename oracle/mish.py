"""ORACLE - test infrastructure only (imported by tests/; never by the product).

Mish as the reference's `Mish_TRT` plugin computes it (yolov4/mish.cu:111-135), restated in NumPy float32:

    tanh_activate_kernel(y) = 2 / (1 + expf(-2 y)) - 1                                   mish.cu:111
    softplus_kernel(x, threshold = 20) = x        if x >  threshold                       mish.cu:113-117
                                         expf(x)  if x < -threshold
                                         logf(expf(x) + 1) otherwise
    mish_kernel: out = x * tanh_activate_kernel(softplus_kernel(x))                      mish.cu:119-135

Pinned on the reference's own kernel (oracle/_ref/libref_yolov4_plugin.so = yolov4/mish.cu compiled by hipcc, run on the MI355X):
tests/golden/ref_plugins.npz case "mish_*" and tests/test_ref_pinning.py.  NumPy's expf / logf and the device's differ by an ulp,
so the pin is at rtol 2e-6; the product's kernel against the reference's kernel is bit-exact (same device math library).
"""
import numpy as np


def mish(x):
    x = np.asarray(x, dtype=np.float32)
    one, two = np.float32(1.0), np.float32(2.0)
    with np.errstate(over="ignore"):
        e = np.exp(x, dtype=np.float32)
        sp = np.where(x > np.float32(20.0), x, np.where(x < np.float32(-20.0), e, np.log(e + one, dtype=np.float32)))
        th = two / (one + np.exp(-two * sp, dtype=np.float32)) - one
    return (x * th).astype(np.float32)


def mish_torch(x):
    """the same expression on a torch fp32 tensor (for the conv -> BN -> Mish_TRT engine tests)"""
    import torch
    e = torch.exp(x)
    sp = torch.where(x > 20.0, x, torch.where(x < -20.0, e, torch.log(e + 1.0)))
    return x * (2.0 / (1.0 + torch.exp(-2.0 * sp)) - 1.0)
