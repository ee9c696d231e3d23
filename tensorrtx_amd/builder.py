"""ctypes binding of the network-definition half of the C ABI (include/trtx_hip.h section 2: createInferBuilder /
createNetworkV2 / INetworkDefinition::add* / buildSerializedNetwork).  The C++ host builders go through include/NvInfer.h;
this thin Python spelling exists for tests and tools that assemble small ad-hoc networks."""
import ctypes

import numpy as np

from .capi import check, lib
from .engine import Dims

P_STRIDE, P_PADDING, P_DILATION, P_GROUPS, P_KERNEL, P_NB_OUT = 1, 2, 3, 4, 15, 16
ACT = {"relu": 0, "sigmoid": 1, "tanh": 2, "leaky": 3}
FLAG_FP16, FLAG_INT8 = 0, 1


def _dims(shape):
    d = Dims()
    d.nb = len(shape)
    for i, v in enumerate(shape):
        d.d[i] = int(v)
    return d


def _f(a):
    a = None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    return a, (a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None), (0 if a is None else a.size)


class Network:
    def __init__(self, max_batch=1, fp16=False, int8=False, explicit_batch=False):
        L = lib()
        self.L = L
        self.b = ctypes.c_void_p()
        check(L.trtx_builder_create(ctypes.byref(self.b)), "trtx_builder_create")
        check(L.trtx_builder_set_max_batch(self.b, max_batch), "set_max_batch")
        if fp16:
            check(L.trtx_builder_set_flag(self.b, FLAG_FP16, 1), "set_flag fp16")
        if int8:
            check(L.trtx_builder_set_flag(self.b, FLAG_INT8, 1), "set_flag int8")
        self.n = ctypes.c_void_p()
        check(L.trtx_network_create(self.b, 1 if explicit_batch else 0, ctypes.byref(self.n)), "trtx_network_create")
        self._keep = []

    def _layer(self, idx, what):
        if idx < 0:
            self.L.trtx_network_last_error.restype = ctypes.c_char_p
            raise RuntimeError(f"{what}: {self.L.trtx_network_last_error(self.n)}")
        return idx

    def out(self, layer, i=0):
        return self.L.trtx_layer_output(self.n, layer, i)

    def input(self, name, shape):
        d = _dims(shape)
        return self._layer(self.L.trtx_add_input(self.n, name.encode(), 0, ctypes.byref(d)), "add_input")

    def _set2(self, layer, param, v):
        arr = (ctypes.c_int32 * 2)(v, v) if np.isscalar(v) else (ctypes.c_int32 * 2)(*v)
        check(self.L.trtx_layer_set_ints(self.n, layer, param, arr, 2), "trtx_layer_set_ints")

    def conv(self, x, w, bias=None, stride=1, padding=0, deconv=False):
        """w: KCRS (conv) or CKRS (deconv) fp32 numpy; returns the layer index"""
        w = np.ascontiguousarray(w, dtype=np.float32)
        nb_out = w.shape[1] if deconv else w.shape[0]
        wa, wp, wn = _f(w)
        ba, bp, bn = _f(bias)
        self._keep += [wa, ba]
        fn = self.L.trtx_add_deconvolution if deconv else self.L.trtx_add_convolution
        l = self._layer(fn(self.n, x, nb_out, w.shape[2], w.shape[3], wp, ctypes.c_int64(wn), bp, ctypes.c_int64(bn)), "add_conv")
        self._set2(l, P_STRIDE, stride)
        self._set2(l, P_PADDING, padding)
        return l

    def activation(self, x, kind):
        return self._layer(self.L.trtx_add_activation(self.n, x, ACT[kind]), "add_activation")

    def scale(self, x, shift, scale, power=None):
        """IScaleLayer, ScaleMode::kCHANNEL (a folded BatchNorm, e.g. addBatchNorm2d of yolov4/yolov4.cpp:181-197)"""
        sa, sp, sn = _f(shift)
        ca, cp, cn = _f(scale)
        pa, pp, pn = _f(np.ones_like(sa) if power is None else power)
        self._keep += [sa, ca, pa]
        return self._layer(self.L.trtx_add_scale(self.n, x, 1, sp, ctypes.c_int64(sn), cp, ctypes.c_int64(cn), pp, ctypes.c_int64(pn)), "add_scale")

    def plugin(self, inputs, name, version="1"):
        """getPluginRegistry()->getPluginCreator(name, version)->createPlugin(name, empty field collection), then addPluginV2 -
        what convBnMish does for "Mish_TRT" (yolov4/yolov4.cpp:207-212).  The v-tables are opaque blobs here (sized generously)."""
        creator = (ctypes.c_void_p * 8)()
        self.L.trtx_registry_get.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
        self.L.trtx_add_plugin_v2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        check(self.L.trtx_registry_get(name.encode(), version.encode(), ctypes.cast(creator, ctypes.c_void_p)), f"no plugin creator {name}/{version}")
        create = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p)(creator[3])
        vt = (ctypes.c_void_p * 16)()
        if create(creator[0], name.encode(), None, 0, ctypes.cast(vt, ctypes.c_void_p)) != 0:
            raise RuntimeError(f"createPlugin({name}) failed")
        arr = (ctypes.c_int32 * len(inputs))(*inputs)
        l = self._layer(self.L.trtx_add_plugin_v2(self.n, ctypes.cast(arr, ctypes.c_void_p), len(inputs), ctypes.cast(vt, ctypes.c_void_p)), "add_plugin_v2")
        destroy = ctypes.CFUNCTYPE(None, ctypes.c_void_p)(vt[13])   # the network holds its own clone
        destroy(vt[0])
        return l

    def pooling(self, x, k, stride, padding=0, avg=False):
        l = self._layer(self.L.trtx_add_pooling(self.n, x, 1 if avg else 0, k, k), "add_pooling")
        self._set2(l, P_STRIDE, stride)
        self._set2(l, P_PADDING, padding)
        return l

    def elementwise(self, a, b, op=0):
        return self._layer(self.L.trtx_add_elementwise(self.n, a, b, op), "add_elementwise")

    def resize_nearest(self, x, scale=2):
        """IResizeLayer, nearest, integer scale on H and W (setScales({1, s, s}))"""
        l = self._layer(self.L.trtx_add_resize(self.n, x), "add_resize")
        sc = (ctypes.c_float * 3)(1.0, float(scale), float(scale))
        check(self.L.trtx_layer_set_floats(self.n, l, 12, sc, 3), "trtx_layer_set_floats(resize scales)")   # TRTX_P_RESIZE_SCALES
        return l

    def slice_channels(self, x, start, size, chw):
        """ISliceLayer over the channel axis of a (C, H, W) tensor: channels [start, start + size) (block.cpp:134-149, the C2f split)"""
        c, h, w = chw
        st, sz, sp = _dims((start, 0, 0)), _dims((size, h, w)), _dims((1, 1, 1))
        return self._layer(self.L.trtx_add_slice(self.n, x, ctypes.byref(st), ctypes.byref(sz), ctypes.byref(sp)), "add_slice")

    def concat(self, tensors):
        arr = (ctypes.c_int32 * len(tensors))(*tensors)
        return self._layer(self.L.trtx_add_concatenation(self.n, arr, len(tensors)), "add_concatenation")

    def mark_output(self, tensor, name):
        check(self.L.trtx_tensor_set_name(self.n, tensor, name.encode()), "set_name")
        check(self.L.trtx_mark_output(self.n, tensor), "mark_output")

    def set_int8_calibrator(self, cal):
        """IBuilderConfig::setInt8Calibrator: cal is a tensorrtx_amd.calibrator.Calibrator (kept alive by this object)"""
        self._keep.append(cal)
        check(self.L.trtx_builder_set_int8_calibrator(self.b, ctypes.byref(cal.vtbl)), "trtx_builder_set_int8_calibrator")

    def build(self):
        hm = ctypes.c_void_p()
        check(self.L.trtx_build_serialized(self.b, self.n, ctypes.byref(hm)), "trtx_build_serialized")
        self.L.trtx_hostmem_data.restype = ctypes.c_void_p
        self.L.trtx_hostmem_size.restype = ctypes.c_size_t
        plan = ctypes.string_at(self.L.trtx_hostmem_data(hm), self.L.trtx_hostmem_size(hm))
        self.L.trtx_hostmem_destroy(hm)
        return plan

    def close(self):
        if self.n:
            self.L.trtx_network_destroy(self.n)
            self.n = None
        if self.b:
            self.L.trtx_builder_destroy(self.b)
            self.b = None
