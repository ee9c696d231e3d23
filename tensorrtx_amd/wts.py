"""`.wts` writer — host-side mirror of the reference's gen_wts.py exporters
(lenet/gen_wts.py:83-92, yolov8/gen_wts.py:50-58; format: tutorials/getting_started.md:107-132).

    <number of blobs>\n
    <name> <count> <hex> <hex> ...\n        each <hex> = struct.pack('>f', v).hex()  (big-endian IEEE-754 bits)

Two dialects exist upstream (single space everywhere, or a double space after the count); both are
produced here so the loader's whitespace handling can be tested.
"""
import numpy as np


def write_wts(path, tensors, dialect="single"):
    """tensors: ordered mapping name -> array-like (any shape, flattened row-major as fp32)."""
    sep_after_count = " " if dialect == "single" else "  "
    with open(path, "wb") as f:
        f.write(f"{len(tensors)}\n".encode())
        for name, arr in tensors.items():
            a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32).reshape(-1))
            f.write(f"{name} {a.size}".encode())
            if a.size:
                # big-endian bytes -> hex digits -> rows of " xxxxxxxx" (fully vectorised: no per-value Python work)
                digits = np.frombuffer(a.astype(">f4").tobytes().hex().encode(), dtype=np.uint8).reshape(-1, 8)
                rows = np.full((a.size, 9), ord(" "), dtype=np.uint8)
                rows[:, 1:] = digits
                f.write(sep_after_count[1:].encode())  # the extra blank of the "double" dialect
                f.write(rows.tobytes())
            f.write(b"\n")
