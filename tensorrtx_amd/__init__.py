"""tensorrtx_amd — MI355X (gfx950) native runtime behind the tensorrtx builder / plugin surface.

The compute lives in ``tensorrtx_amd/lib/libtrtx_hip.so`` (hand-written HIP, C ABI declared in
``include/trtx_hip.h``).  This Python package is a thin ctypes binding used by the tests and by
``bench.py``; PyTorch is used only for device memory, streams and ``torch.distributed``.
There is no CPU fallback: every entry point raises if the HIP library or a GPU is missing.
"""
from .capi import TrtxError, lib, lib_path  # noqa: F401

__all__ = ["TrtxError", "lib", "lib_path"]
