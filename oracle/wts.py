"""ORACLE (test infrastructure): pure-Python restatement of the reference's loadWeights
(lenet/utils.h:49-80; yolov8/src/block.cpp:13-43): `istream >> count`, then per blob
`>> name >> dec >> size` and `size` times `>> hex >> uint32` reinterpreted as fp32.
`>>` skips any whitespace, so the reader is whitespace-agnostic."""
import binascii
import re

import numpy as np

_TOKEN = re.compile(rb"\S+")
_WS = frozenset(b" \t\r\n\v\f")


def load_wts(path):
    with open(path, "rb") as f:
        data = f.read()
    view = np.frombuffer(data, dtype=np.uint8)

    def token(pos):
        m = _TOKEN.search(data, pos)
        assert m, "unexpected end of file"
        return m.group(), m.end()

    t, pos = token(0)
    count = int(t)
    assert count > 0, "Invalid weight map file."
    out = {}
    for _ in range(count):
        name, pos = token(pos)
        t, pos = token(pos)
        size = int(t)
        bits = None
        if size:
            # fast path for what the exporters write: `size` full 8-digit words separated by single blanks
            start = _TOKEN.search(data, pos).start()
            end = start + 9 * size - 1
            if end < len(data) and data[end] in _WS:
                words = view[start:end + 1].reshape(size, 9)  # zero-copy: 8 digits + the separator that follows
                sep, digits = words[:, 8], words[:, :8]
                # ASCII whitespace is {9..13, 32}; hex digits are all > 32
                if digits.min() > 32 and sep.max() <= 32 and sep.min() >= 9 and not ((sep > 13) & (sep < 32)).any():
                    bits = np.frombuffer(binascii.unhexlify(digits.tobytes()), dtype=">u4").astype(np.uint32)
                    pos = end
        if bits is None:  # general path: any token width, any whitespace
            vals = []
            for _k in range(size):
                t, pos = token(pos)
                vals.append(int(t, 16))
            bits = np.array(vals, dtype=np.uint32)
        out[name.decode()] = bits.view(np.float32)
    return out
