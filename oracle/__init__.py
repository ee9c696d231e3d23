"""ORACLE — test infrastructure only.

CPU restatements of the wang-xinyu/tensorrtx algorithms on the hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product (``tensorrtx_amd``) never imports it.

Parity status: the `.wts` reader and the LeNet restatement are pinned on fixtures generated from the
reference's own runnable Python (tests/golden/make_golden.py imports lenet/gen_wts.py from /root/reference);
everything else is **parity unpinned** — the reference ships no golden vectors, unit tests or weights
for this path and cannot be compiled here (TensorRT / CUDA / OpenCV absent), so
oracle/_ref does not exist; each function cites the reference file:line it restates.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile the plain-C restatements (gcc) into oracle/_build/liboracle.so."""
    so = os.path.join(_HERE, "_build", "liboracle.so")
    srcs = [os.path.join(_HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(_HERE, "csrc")))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "_build/liboracle.so"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB
