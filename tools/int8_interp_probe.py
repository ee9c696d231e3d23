"""Small ad-hoc kINT8 networks on the GPU against oracle/lowered_int8.py (the plan interpreted on the CPU at its own scales): which construct, if any, makes an
int8 engine deviate from what its plan says.  Scales are fabricated (a calibration cache with one value for every tensor): nothing here depends on a calibrator."""
import os
import struct
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lowered_int8 as li  # noqa: E402
from tensorrtx_amd import builder, calibrator, engine  # noqa: E402


def bn(rng, c):
    return rng.normal(0, 0.3, c).astype(np.float32), rng.uniform(0.7, 1.3, c).astype(np.float32)


def cbs(net, rng, x, cin, cout, k=3, s=1, act="silu"):
    w = rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, (cout, cin, k, k)).astype(np.float32)
    t = net.out(net.conv(x, w, stride=s, padding=k // 2))
    sh, sc = bn(rng, cout)
    t = net.out(net.scale(t, sh, sc))
    if act == "silu":
        sg = net.out(net.activation(t, "sigmoid"))
        t = net.out(net.elementwise(t, sg, 1))
    elif act == "relu":
        t = net.out(net.activation(t, "relu"))
    return t


CASES = {}


def case(f):
    CASES[f.__name__] = f
    return f


@case
def plain_chain(net, rng, x):
    a = cbs(net, rng, x, 16, 32)
    b = cbs(net, rng, a, 32, 32)
    c = cbs(net, rng, b, 32, 64, k=1)
    return cbs(net, rng, c, 64, 32, k=3, act="none")


@case
def shortcut(net, rng, x):
    a = cbs(net, rng, x, 16, 32)
    b = cbs(net, rng, cbs(net, rng, a, 32, 32), 32, 32)
    y = net.out(net.elementwise(a, b, 0))
    return cbs(net, rng, y, 32, 32, k=1, act="none")


@case
def relu_shortcut(net, rng, x):
    a = cbs(net, rng, x, 16, 64, act="relu")
    b = cbs(net, rng, cbs(net, rng, a, 64, 64, k=1, act="relu"), 64, 64, act="none")
    y = net.out(net.activation(net.out(net.elementwise(b, a, 0)), "relu"))
    return cbs(net, rng, y, 64, 32, k=1, act="none")


@case
def sppf(net, rng, x):
    a = cbs(net, rng, cbs(net, rng, x, 16, 64), 64, 32, k=1)
    p1 = net.out(net.pooling(a, 5, 1, 2))
    p2 = net.out(net.pooling(p1, 5, 1, 2))
    p3 = net.out(net.pooling(p2, 5, 1, 2))
    return cbs(net, rng, net.out(net.concat([a, p1, p2, p3])), 128, 64, k=1, act="none")


@case
def upsample_concat(net, rng, x):
    skip = cbs(net, rng, x, 16, 32)
    low = cbs(net, rng, skip, 32, 64, s=2)
    low = cbs(net, rng, low, 64, 64, k=1)
    up = net.out(net.resize_nearest(low, 2))
    return cbs(net, rng, cbs(net, rng, net.out(net.concat([up, skip])), 96, 64, k=1), 64, 32, act="none")


@case
def maxpool_between(net, rng, x):
    a = cbs(net, rng, x, 16, 64, act="relu")
    p = net.out(net.pooling(a, 3, 2, 1))
    return cbs(net, rng, cbs(net, rng, p, 64, 64, k=1, act="relu"), 64, 32, act="none")


@case
def c2f(net, rng, x):
    # yolov8/src/block.cpp:112-155: cv1 -> split in two halves -> bottleneck(s) on the second half with shortcut -> concat [half0, half1, m0] -> cv2
    a = cbs(net, rng, x, 16, 64, k=1)
    h1 = net.out(net.slice_channels(a, 32, 32, (64, 40, 40)))
    m = cbs(net, rng, cbs(net, rng, h1, 32, 32), 32, 32)
    m = net.out(net.elementwise(h1, m, 0))
    return cbs(net, rng, net.out(net.concat([a, m])), 96, 64, k=1, act="none")


@case
def c2f_m_only(net, rng, x):
    a = cbs(net, rng, x, 16, 64, k=1)
    h1 = net.out(net.slice_channels(a, 32, 32, (64, 40, 40)))
    m = cbs(net, rng, cbs(net, rng, h1, 32, 32), 32, 32)
    m = net.out(net.elementwise(h1, m, 0))
    return cbs(net, rng, m, 32, 64, k=1, act="none")


@case
def c2f_no_shortcut(net, rng, x):
    a = cbs(net, rng, x, 16, 64, k=1)
    h1 = net.out(net.slice_channels(a, 32, 32, (64, 40, 40)))
    m = cbs(net, rng, cbs(net, rng, h1, 32, 32), 32, 32)
    return cbs(net, rng, net.out(net.concat([a, m])), 96, 64, k=1, act="none")


@case
def slice_then_chain(net, rng, x):
    a = cbs(net, rng, x, 16, 64, k=1)
    h1 = net.out(net.slice_channels(a, 32, 32, (64, 40, 40)))
    return cbs(net, rng, cbs(net, rng, h1, 32, 32), 32, 32, act="none")


@case
def fp16_3x3_to_int8(net, rng, x):
    a = cbs(net, rng, x, 16, 32, k=1)          # K = 16: stays an fp16 tensor
    return cbs(net, rng, cbs(net, rng, a, 32, 32), 32, 32, act="none")


@case
def c2f_int8(net, rng, x):
    # the real thing (block.cpp:112-155) with an int8 concat buffer: cv1 has K = 64
    x64 = cbs(net, rng, x, 16, 64)
    a = cbs(net, rng, x64, 64, 64, k=1)
    h1 = net.out(net.slice_channels(a, 32, 32, (64, 40, 40)))
    m = cbs(net, rng, cbs(net, rng, h1, 32, 32), 32, 32)
    m = net.out(net.elementwise(h1, m, 0))
    m2 = cbs(net, rng, cbs(net, rng, m, 32, 32), 32, 32)
    m2 = net.out(net.elementwise(m, m2, 0))
    return cbs(net, rng, net.out(net.concat([a, m, m2])), 128, 64, k=1, act="none")


@case
def stem3(net, rng, x3):
    # the 3-channel stem kernel in front of an int8 layer (model.cpp:115-118)
    a = cbs(net, rng, x3, 3, 16, s=2)
    b = cbs(net, rng, a, 16, 32, s=2)
    return cbs(net, rng, cbs(net, rng, b, 32, 32), 32, 32, k=1, act="none")


@case
def head_arms(net, rng, x):
    # two arms reading ONE int8 tensor, 64- and 80-channel outputs concatenated (the detect head's cv2 / cv3, model.cpp:188-251)
    a = cbs(net, rng, cbs(net, rng, x, 16, 64), 64, 64)
    b = cbs(net, rng, cbs(net, rng, a, 64, 64), 64, 64, k=1, act="none")
    c = cbs(net, rng, cbs(net, rng, a, 64, 80), 80, 80, k=1, act="none")
    return net.out(net.concat([b, c]))


def run(name, scale=0.04, B=2, H=40, W=40):
    rng = np.random.default_rng(abs(hash(name)) % 1000)
    plans = {}
    for int8 in (0, 1):
        net = builder.Network(max_batch=B, fp16=True, int8=bool(int8))
        cin0 = 3 if name == "stem3" else 16
        x = net.input("data", (cin0, H, W))
        rng = np.random.default_rng(7)
        net.mark_output(CASES[name](net, rng, x), "y")
        if int8:
            names = [t["name"] or f"(Unnamed Tensor* {t['id']})" for t in engine.describe_plan(plans[0])["tensors"]]
            # a different scale for every tensor (0.6 .. 1.6 x `scale`): concat buffers re-scale their producers, the int8 resize requantises
            cache = b"TRT-8601-EntropyCalibration2\n" + b"".join(f"{n}: {struct.unpack('<I', struct.pack('<f', scale * (0.6 + 0.1 * ((7 * i) % 11))))[0]:08x}\n".encode()
                                                                 for i, n in enumerate(names))
            net.set_int8_calibrator(calibrator.Calibrator(cache=cache))
            plans[1] = net.build()
        else:
            plans[0] = net.build()
        net.close()
    plan = plans[1]
    low = engine.describe_plan(plan, lowered=True)
    xin = np.random.default_rng(3).normal(0, 1, (B, cin0, H, W)).astype(np.float32)
    emu = li.run(plan, engine.describe_plan(plan), low, {"data": xin}, B)["y"].reshape(B, -1)
    if not torch.cuda.is_available():
        print(f"{name}: interpreter ran ({emu.shape}, range +-{np.abs(emu).max():.3g}); ops {[o['kind'] for o in low['ops']]}")
        return 0.0, 0.0
    e = engine.Engine(plan)
    dev = torch.device("cuda:0")
    bufs = [torch.from_numpy(xin).to(dev)] + [torch.zeros(B * int(np.prod(e.dims[i])), dtype=torch.float32, device=dev) for i in range(1, e.nb_bindings)]
    e.enqueue(B, bufs)
    torch.cuda.synchronize()
    got = bufs[1].cpu().numpy().reshape(B, -1)
    e.close()
    d = np.abs(got - emu)
    i8 = [(o["i8"], o["name"][:24]) for o in low["ops"] if o["kind"] == "conv"]
    print(f"{name:18s} ops {[o['kind'] for o in low['ops']]}\n   conv int8 flags {[f for f, _ in i8]}\n   |engine - interpreter| max {d.max():.4g} mean {d.mean():.3g}  "
          f"output range +-{np.abs(emu).max():.3g}; values differing by > 1e-3: {(d > 1e-3).mean():.4f}")
    return float(d.max()), float((d > 1e-3).mean())


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CASES)):
        run(n)
