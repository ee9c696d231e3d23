"""ORACLE (test infrastructure): pure-Python restatement of the reference's loadWeights
(lenet/utils.h:49-80; yolov8/src/block.cpp:13-43): `istream >> count`, then per blob
`>> name >> dec >> size` and `size` times `>> hex >> uint32` reinterpreted as fp32.
`>>` skips any whitespace, so the reader is whitespace-agnostic."""
import numpy as np


def load_wts(path):
    with open(path, "r") as f:
        tok = f.read().split()
    pos = 0
    count = int(tok[pos]); pos += 1
    assert count > 0, "Invalid weight map file."
    out = {}
    for _ in range(count):
        name = tok[pos]; size = int(tok[pos + 1]); pos += 2
        bits = np.array([int(t, 16) for t in tok[pos:pos + size]], dtype=np.uint32)
        pos += size
        out[name] = bits.view(np.float32)
    return out
