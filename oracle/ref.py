"""ORACLE — test infrastructure only.  Python access to ``oracle/_ref/``: the reference's OWN code compiled from
/root/reference by ``oracle/ref_build.py`` (see there for what is built and how).

* ``host()`` + ``yolov8_nms`` / ``yolov5_nms`` / ``retina_nms``: the reference's host post-processing (CPU).
* ``load_plugins(family)`` + ``run_plugin``: the reference's CUDA plugins, built by hipcc as *user plugins* against
  include/NvInfer.h.  dlopening a family registers its creators with libtrtx_hip.so's registry (later registration
  wins); ``load_plugins`` captures the reference creators and then restores whatever was registered before, so the
  product's built-in plugins keep serving every other test in the process.  ``run_plugin`` drives a creator through the
  C-ABI v-table exactly as the engine does: deserialize/create -> initialize -> getWorkspaceSize -> enqueue -> terminate.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_HOST = None
_FAMILIES = {}


def available(name="libref_host.so"):
    return os.path.exists(os.path.join(_REF, name))


def _need(name):
    p = os.path.join(_REF, name)
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} is missing: run `python oracle/ref_build.py` in the build container (needs /root/reference)")
    return p


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def host():
    global _HOST
    if _HOST is None:
        _HOST = ctypes.CDLL(_need("libref_host.so"))
    return _HOST


def yolov8_nms(output_row, conf_thresh=0.5, nms_thresh=0.45, cap=4096):
    """nms() of yolov8/src/postprocess.cpp:94-121 on one image's decode buffer -> kept detections [n, 90], emission order."""
    L = host()
    df = L.ref_yolov8_det_floats()
    row = np.ascontiguousarray(output_row, dtype=np.float32).copy()
    out = np.zeros((cap, df), dtype=np.float32)
    n = L.ref_yolov8_nms(_fp(row), ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh), _fp(out), cap)
    assert n <= cap
    return out[:n]


def yolov8_batch_nms(output, conf_thresh=0.5, nms_thresh=0.45, cap=1024):
    """batch_nms() of yolov8/src/postprocess.cpp:123-129 -> list of [n_b, 90]."""
    L = host()
    df = L.ref_yolov8_det_floats()
    o = np.ascontiguousarray(output, dtype=np.float32).copy()
    B = o.shape[0]
    out = np.zeros((B, cap, df), dtype=np.float32)
    cnt = np.zeros(B, dtype=np.int32)
    L.ref_yolov8_batch_nms(_fp(o), B, o.shape[1], ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh), _fp(out), cap,
                           cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return [out[b, :min(cnt[b], cap)] for b in range(B)]


def yolov8_nms_obb(output_row, conf_thresh=0.5, nms_thresh=0.45, cap=4096):
    """nms_obb() of yolov8/src/postprocess.cpp:357-385 (ProbIoU) -> kept detections [n, 90]."""
    L = host()
    row = np.ascontiguousarray(output_row, dtype=np.float32).copy()
    out = np.zeros((cap, 90), dtype=np.float32)
    n = L.ref_yolov8_nms_obb(_fp(row), ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh), _fp(out), cap)
    return out[:n]


def yolov5_nms(output_row, conf_thresh=0.5, nms_thresh=0.45, cap=4096):
    """nms() of yolov5/src/postprocess.cpp:50-73 (centre-format boxes) -> kept detections [n, 38]."""
    L = host()
    df = L.ref_yolov5_det_floats()
    row = np.ascontiguousarray(output_row, dtype=np.float32).copy()
    out = np.zeros((cap, df), dtype=np.float32)
    n = L.ref_yolov5_nms(_fp(row), ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh), _fp(out), cap)
    return out[:n]


def retina_nms(output_row, nms_thresh=0.4, cap=8192):
    """nms() of retinaface/common.hpp:110-130 -> kept detections [n, 15]."""
    L = host()
    row = np.ascontiguousarray(output_row, dtype=np.float32).copy()
    out = np.zeros((cap, 15), dtype=np.float32)
    n = L.ref_retina_nms(_fp(row), ctypes.c_float(nms_thresh), _fp(out), cap)
    assert n <= cap
    return out[:n]


# ------------------------------------------------------------------------------------------------ GPU plugins
class Dims(ctypes.Structure):
    _fields_ = [("nb", ctypes.c_int32), ("d", ctypes.c_int64 * 8)]


class PluginVtbl(ctypes.Structure):
    pass


_V = ctypes.c_void_p
PluginVtbl._fields_ = [
    ("self", _V),
    ("get_nb_outputs", ctypes.CFUNCTYPE(ctypes.c_int32, _V)),
    ("get_output_dims", ctypes.CFUNCTYPE(ctypes.c_int32, _V, ctypes.c_int32, ctypes.POINTER(Dims), ctypes.c_int32, ctypes.POINTER(Dims))),
    ("configure", ctypes.CFUNCTYPE(ctypes.c_int32, _V, ctypes.POINTER(Dims), ctypes.c_int32, ctypes.POINTER(Dims), ctypes.c_int32, ctypes.c_int32)),
    ("initialize", ctypes.CFUNCTYPE(ctypes.c_int32, _V)),
    ("terminate", ctypes.CFUNCTYPE(None, _V)),
    ("workspace_size", ctypes.CFUNCTYPE(ctypes.c_size_t, _V, ctypes.c_int32)),
    ("enqueue", ctypes.CFUNCTYPE(ctypes.c_int32, _V, ctypes.c_int32, ctypes.POINTER(_V), ctypes.POINTER(_V), _V, _V)),
    ("serialization_size", ctypes.CFUNCTYPE(ctypes.c_size_t, _V)),
    ("serialize", ctypes.CFUNCTYPE(None, _V, _V)),
    ("plugin_type", ctypes.CFUNCTYPE(ctypes.c_char_p, _V)),
    ("plugin_version", ctypes.CFUNCTYPE(ctypes.c_char_p, _V)),
    ("clone", ctypes.CFUNCTYPE(ctypes.c_int32, _V, ctypes.POINTER(PluginVtbl))),
    ("destroy", ctypes.CFUNCTYPE(None, _V)),
]


class PluginField(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", _V), ("type", ctypes.c_int32), ("length", ctypes.c_int32)]


class CreatorVtbl(ctypes.Structure):
    _fields_ = [
        ("self", _V),
        ("plugin_name", ctypes.CFUNCTYPE(ctypes.c_char_p, _V)),
        ("plugin_version", ctypes.CFUNCTYPE(ctypes.c_char_p, _V)),
        ("create", ctypes.CFUNCTYPE(ctypes.c_int32, _V, ctypes.c_char_p, ctypes.POINTER(PluginField), ctypes.c_int32, ctypes.POINTER(PluginVtbl))),
        ("deserialize", ctypes.CFUNCTYPE(ctypes.c_int32, _V, ctypes.c_char_p, _V, ctypes.c_size_t, ctypes.POINTER(PluginVtbl))),
    ]


FAMILY_PLUGINS = {
    "yolov8_plugin": ["YoloLayer_TRT"],
    "yolov5_plugin": ["YoloLayer_TRT"],
    "retinaface_plugin": ["Decode_TRT"],
    "yolov4_plugin": ["Mish_TRT"],
    "rcnn_plugins": ["RpnDecode", "RpnNms", "RoiAlign", "PredictorDecode", "BatchedNms", "MaskRcnnInference"],
    "yolov8_post": [],
}


def _trtx():
    import tensorrtx_amd
    L = tensorrtx_amd.lib()
    L.trtx_registry_get.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(CreatorVtbl)]
    L.trtx_registry_register.argtypes = [ctypes.POINTER(CreatorVtbl)]
    return L


def registry_get(name, version="1"):
    c = CreatorVtbl()
    return c if _trtx().trtx_registry_get(name.encode(), version.encode(), ctypes.byref(c)) == 0 else None


def registry_register(creator):
    assert _trtx().trtx_registry_register(ctypes.byref(creator)) == 0


def load_plugins(family):
    """dlopen oracle/_ref/libref_<family>.so; returns {plugin name: reference CreatorVtbl}.  The registry is left as found."""
    if family in _FAMILIES:
        return _FAMILIES[family][1]
    L = _trtx()
    names = FAMILY_PLUGINS[family]
    before = {n: registry_get(n) for n in names}
    lib = ctypes.CDLL(_need(f"libref_{family}.so"), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    ref = {}
    for n in names:
        c = registry_get(n)
        assert c is not None and (before[n] is None or c.self != before[n].self), f"reference creator {n} did not register"
        ref[n] = c
        if before[n] is not None:
            registry_register(before[n])  # the product's plugin serves the rest of the process again
    _FAMILIES[family] = (lib, ref)
    del L
    return ref


def family_lib(family):
    load_plugins(family)
    return _FAMILIES[family][0]


class use_creator:
    """Context manager: make `creator` the registered one for its (name, version) — as an application that links the
    reference's plugin would — and put the previous registration back afterwards."""

    def __init__(self, creator):
        self.c = creator
        self.name = creator.plugin_name(creator.self).decode()
        self.prev = None

    def __enter__(self):
        self.prev = registry_get(self.name)
        registry_register(self.c)
        return self

    def __exit__(self, *a):
        if self.prev is not None:
            registry_register(self.prev)


def make_plugin(creator, blob=None, fields=None):
    """deserializePlugin(blob) or createPlugin(fields = [(name, np.ndarray int32/float32)]) through the C ABI."""
    v = PluginVtbl()
    name = creator.plugin_name(creator.self)
    if blob is not None:
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        rc = creator.deserialize(creator.self, name, ctypes.cast(buf, _V), len(blob), ctypes.byref(v))
    else:
        # a field is (name, array) or (name, array, length) when the creator counts elements of a struct type (e.g. YoloKernel)
        keep = [np.ascontiguousarray(f[1]) for f in fields]
        arr = (PluginField * max(len(fields), 1))()
        for i, (f, a) in enumerate(zip(fields, keep)):
            arr[i] = PluginField(f[0].encode(), a.ctypes.data_as(_V), 5 if a.dtype == np.int32 else (1 if a.dtype == np.float32 else 8),
                                 f[2] if len(f) > 2 else a.size)
        rc = creator.create(creator.self, name, arr, len(fields), ctypes.byref(v))
    assert rc == 0, "plugin construction failed"
    return v


def plugin_blob(v):
    n = v.serialization_size(v.self)
    buf = ctypes.create_string_buffer(n)
    v.serialize(v.self, ctypes.cast(buf, _V))
    return buf.raw


def run_plugin(v, batch, inputs, out_shapes, destroy=True):
    """initialize -> workspace -> enqueue -> terminate on CUDA fp32 tensors; returns the output tensors."""
    import torch
    dev = inputs[0].device
    ins = [t.contiguous() for t in inputs]
    outs = [torch.zeros(s, dtype=torch.float32, device=dev) for s in out_shapes]
    assert v.initialize(v.self) == 0
    ws_bytes = v.workspace_size(v.self, batch)
    ws = torch.zeros(max(int(ws_bytes), 16), dtype=torch.uint8, device=dev)
    ip = (_V * len(ins))(*[t.data_ptr() for t in ins])
    op = (_V * len(outs))(*[t.data_ptr() for t in outs])
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    rc = v.enqueue(v.self, batch, ip, op, _V(ws.data_ptr()), _V(st))
    torch.cuda.synchronize()
    assert rc == 0, f"plugin enqueue returned {rc}"
    v.terminate(v.self)
    if destroy:
        v.destroy(v.self)
    return outs
