"""INT8 calibrator for tests / tools: Python implementation of the C calibrator v-table (include/trtx_hip.h, the
IInt8EntropyCalibrator2 methods of the reference's yolov8/include/calibrator.h:14-36).

    cal = Calibrator(batches=[x0, x1, ...], batch_size=B)        # CUDA fp32 tensors, one per calibration batch
    cal = Calibrator(cache=open("int8calib.table", "rb").read()) # scales from a cache: the build needs no GPU
    with cal.installed():                                        # used by engine.build_plan(..., int8=1)
        plan = engine.build_plan("yolov8n", wts, batch=B, int8=1)
    cal.written_cache                                            # what the builder handed to writeCalibrationCache
"""
import contextlib
import ctypes

from .engine import models_lib


class CalibratorVtbl(ctypes.Structure):
    _fields_ = [
        ("self", ctypes.c_void_p),
        ("get_batch_size", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p)),
        ("get_batch", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32)),
        ("read_cache", ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t))),
        ("write_cache", ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)),
        ("get_algorithm", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p)),
    ]


class Calibrator:
    def __init__(self, batches=None, batch_size=1, cache=None, algorithm="entropy2"):
        self.batches = list(batches or [])
        self.batch_size = batch_size
        self.cache = cache
        self.written_cache = None
        self._i = 0
        self._keep = None
        self._cache_buf = ctypes.create_string_buffer(cache, len(cache)) if cache else None

        def get_batch_size(_):
            return self.batch_size

        def get_batch(_, bindings, names, nb):
            if self._i >= len(self.batches):
                return 0
            b = self.batches[self._i]
            self._i += 1
            self._keep = b
            for k in range(nb):
                t = b[names[k].decode()] if isinstance(b, dict) else b
                bindings[k] = t.data_ptr()
            return 1

        def read_cache(_, length):
            if not self._cache_buf:
                length[0] = 0
                return None
            length[0] = len(self.cache)
            return ctypes.cast(self._cache_buf, ctypes.c_void_p).value

        def write_cache(_, ptr, n):
            self.written_cache = ctypes.string_at(ptr, n)

        algo = {"entropy": 1, "entropy2": 2, "minmax": 3}[algorithm]   # nvinfer1::CalibrationAlgoType

        def get_algorithm(_):
            return algo

        self.vtbl = CalibratorVtbl()
        self._cbs = (CalibratorVtbl._fields_[1][1](get_batch_size), CalibratorVtbl._fields_[2][1](get_batch),
                     CalibratorVtbl._fields_[3][1](read_cache), CalibratorVtbl._fields_[4][1](write_cache),
                     CalibratorVtbl._fields_[5][1](get_algorithm))
        self.vtbl.get_batch_size, self.vtbl.get_batch, self.vtbl.read_cache, self.vtbl.write_cache, self.vtbl.get_algorithm = self._cbs

    @contextlib.contextmanager
    def installed(self):
        L = models_lib()
        L.trtx_host_set_calibrator(ctypes.byref(self.vtbl))
        try:
            yield self
        finally:
            L.trtx_host_set_calibrator(None)


def parse_cache(text):
    """'TRT-...-EntropyCalibration2' text -> {tensor name: scale}"""
    import struct
    out = {}
    for line in text.decode().split("\n")[1:]:
        if ": " in line:
            name, hexbits = line.rsplit(": ", 1)
            out[name] = struct.unpack("<f", struct.pack("<I", int(hexbits, 16)))[0]
    return out
