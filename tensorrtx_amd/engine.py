"""ctypes binding of C ABI section 2 (builder/engine/context) + the C++ host model builders."""
import ctypes
import json
import os

from . import capi
from .capi import check, lib

_HERE = os.path.dirname(os.path.abspath(__file__))
_MODELS = None


class Dims(ctypes.Structure):
    _fields_ = [("nb", ctypes.c_int32), ("d", ctypes.c_int64 * 8)]


def models_lib():
    global _MODELS
    if _MODELS is None:
        lib()  # libtrtx_hip first (rpath $ORIGIN resolves it too)
        p = os.path.join(_HERE, "lib", "libtrtx_models.so")
        ab = os.environ.get("TRTX_HIP_LIB")   # an A/B build directory carries its own host builders (they resolve libtrtx_hip.so next to themselves)
        if ab and os.path.exists(os.path.join(os.path.dirname(ab), "libtrtx_models.so")):
            p = os.path.join(os.path.dirname(ab), "libtrtx_models.so")
        if not os.path.exists(p):
            raise ImportError(f"{p} is missing: run __graft_entry__.build()")
        _MODELS = ctypes.CDLL(p)
    return _MODELS


def build_plan(model: str, wts_path: str, **options) -> bytes:
    """Run the C++ host builder (tensorrtx_amd/host/*.cpp) and return the serialized plan. CPU-only work."""
    opts = ";".join(f"{k}={int(v)}" for k, v in options.items())
    blob, size = ctypes.c_void_p(), ctypes.c_size_t()
    check(models_lib().trtx_host_build(model.encode(), wts_path.encode(), opts.encode(), ctypes.byref(blob),
                                       ctypes.byref(size)), f"trtx_host_build({model})")
    data = ctypes.string_at(blob, size.value)
    models_lib().trtx_host_free(blob)
    return data


def describe_plan(plan: bytes, lowered: bool = False) -> dict:
    out = ctypes.c_char_p()
    L = lib()
    check(L.trtx_plan_describe(plan, ctypes.c_size_t(len(plan)), 1 if lowered else 0, ctypes.byref(out)),
          "trtx_plan_describe")
    js = ctypes.string_at(out).decode()
    L.trtx_string_free(out)
    return json.loads(js)


class Engine:
    """IRuntime::deserializeCudaEngine + createExecutionContext.  Needs a GPU (no CPU fallback)."""

    def __init__(self, plan: bytes):
        L = lib()
        L.trtx_engine_binding_name.restype = ctypes.c_char_p
        L.trtx_engine_device_memory.restype = ctypes.c_size_t
        self._plan = plan
        self._e = ctypes.c_void_p()
        check(L.trtx_engine_deserialize(plan, ctypes.c_size_t(len(plan)), ctypes.byref(self._e)),
              "trtx_engine_deserialize")
        self._c = ctypes.c_void_p()
        check(L.trtx_context_create(self._e, ctypes.byref(self._c)), "trtx_context_create")
        self.nb_bindings = L.trtx_engine_nb_bindings(self._e)
        self.max_batch = L.trtx_engine_max_batch(self._e)
        self.names = [L.trtx_engine_binding_name(self._e, i).decode() for i in range(self.nb_bindings)]
        self.is_input = [bool(L.trtx_engine_binding_is_input(self._e, i)) for i in range(self.nb_bindings)]
        self.dims = []
        for i in range(self.nb_bindings):
            d = Dims()
            check(L.trtx_engine_binding_dims(self._e, i, ctypes.byref(d)), "binding_dims")
            self.dims.append([d.d[k] for k in range(d.nb)])
        self.device_memory = L.trtx_engine_device_memory(self._e)
        self.device = L.trtx_engine_device(self._e)  # HIP ordinal the weights / arena live on (current device at creation)

    def enqueue(self, batch, bindings, stream=None):
        """bindings: list of CUDA tensors in binding order (inputs first). Async on the current stream."""
        _enqueue(self._c, self.nb_bindings, batch, bindings, stream)

    def enqueue_frames(self, batch, frames, bindings, stream=None):
        """letterbox + enqueue in one call: the stem samples the uint8 HWC BGR frames itself (trtx_context_enqueue_frames)"""
        _enqueue_frames(self._c, self.nb_bindings, batch, frames, bindings, stream)

    def profile(self, batch, bindings):
        arr = (ctypes.c_void_p * self.nb_bindings)(*[t.data_ptr() for t in bindings])
        out = ctypes.c_char_p()
        check(lib().trtx_context_profile(self._c, batch, arr, capi._stream(), ctypes.byref(out)), "trtx_context_profile")
        js = ctypes.string_at(out).decode()
        lib().trtx_string_free(out)
        return json.loads(js)

    def tactics(self):
        """What the tactic timing at deserializeCudaEngine decided: one record per MFMA convolution (trtx_engine_tactics)."""
        out = ctypes.c_char_p()
        check(lib().trtx_engine_tactics(self._e, ctypes.byref(out)), "trtx_engine_tactics")
        js = ctypes.string_at(out).decode()
        lib().trtx_string_free(out)
        return json.loads(js)

    def create_context(self):
        """ICudaEngine::createExecutionContext: a further context (own activation arena, lane streams and events) over the same
        weights.  Several contexts may be in flight at once, each on its own stream (bench.py --contexts)."""
        return ExecutionContext(self)

    def close(self):
        for c in getattr(self, "_extra", []):
            c.close()
        self._extra = []
        if self._c:
            lib().trtx_context_destroy(self._c)
            self._c = None
        if self._e:
            lib().trtx_engine_destroy(self._e)
            self._e = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _enqueue_frames(ctx, nb, batch, frames, bindings, stream):
    """frames: list of uint8 HWC BGR CUDA tensors (one per image, any sizes); bindings as for enqueue (the input entry may be None)."""
    arr = (ctypes.c_void_p * nb)(*[(t.data_ptr() if t is not None else None) for t in bindings])
    fp = (ctypes.c_void_p * batch)(*[f.data_ptr() for f in frames[:batch]])
    fw = (ctypes.c_int32 * batch)(*[int(f.shape[1]) for f in frames[:batch]])
    fh = (ctypes.c_int32 * batch)(*[int(f.shape[0]) for f in frames[:batch]])
    st = capi._stream() if stream is None else ctypes.c_void_p(stream)
    check(lib().trtx_context_enqueue_frames(ctx, batch, fp, fw, fh, arr, st), "trtx_context_enqueue_frames")


def _enqueue(ctx, nb, batch, bindings, stream):
    arr = (ctypes.c_void_p * nb)(*[t.data_ptr() for t in bindings])
    st = capi._stream() if stream is None else ctypes.c_void_p(stream)
    check(lib().trtx_context_enqueue(ctx, batch, arr, st), "trtx_context_enqueue")


class ExecutionContext:
    """One more IExecutionContext of an Engine (the Engine object owns it and destroys it before the engine)."""

    def __init__(self, eng: Engine):
        self.engine = eng
        self._c = ctypes.c_void_p()
        check(lib().trtx_context_create(eng._e, ctypes.byref(self._c)), "trtx_context_create")
        if not hasattr(eng, "_extra"):
            eng._extra = []
        eng._extra.append(self)

    def enqueue(self, batch, bindings, stream=None):
        _enqueue(self._c, self.engine.nb_bindings, batch, bindings, stream)

    def enqueue_frames(self, batch, frames, bindings, stream=None):
        _enqueue_frames(self._c, self.engine.nb_bindings, batch, frames, bindings, stream)

    def close(self):
        if self._c:
            lib().trtx_context_destroy(self._c)
            self._c = None
