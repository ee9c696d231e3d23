"""ctypes binding of the C ABI in include/trtx_hip.h (section 1: plugin operators + single kernels)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class TrtxError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        msg = lib().trtx_status_string(status).decode() if _LIB is not None else "?"
        super().__init__(f"{what}: status {status} ({msg})")


def lib_path() -> str:
    # TRTX_HIP_LIB: another BUILD of this same library (A/B measurements of kernel changes on one box); never a different implementation
    return os.environ.get("TRTX_HIP_LIB") or os.path.join(_HERE, "lib", "libtrtx_hip.so")


def hip_runtimes_mapped():
    """Paths of every libamdhip64 mapped into this process (must be exactly one once lib() has run)."""
    seen = []
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.rsplit(" ", 1)[-1].strip()
            if "libamdhip64" in os.path.basename(path) and path not in seen:
                seen.append(path)
    return seen


def _bind_hip_runtime():
    """libtrtx_hip.so NEEDs `libamdhip64.so.7` and PyTorch-ROCm bundles its own copy under torch/lib with the
    same SONAME.  Whichever copy is mapped first satisfies the other's NEEDED entry *only* in the torch-first
    order (torch dlopens its copy by path; ours is looked up by SONAME).  Two runtimes in one process each own a
    separate device table: torch sees the GPU, trtx_device_count() in the other does not.  So the binding always
    brings torch's runtime in first; a process that never uses torch (pure C++ callers) links the system ROCm."""
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so; plumbing only)


def lib() -> ctypes.CDLL:
    """Load libtrtx_hip.so.  Fails loudly when the extension has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _bind_hip_runtime()
        L = ctypes.CDLL(p)
        rts = hip_runtimes_mapped()
        if len(rts) > 1:
            raise ImportError("two HIP runtimes are mapped in this process (%s): something dlopened a HIP library "
                              "before tensorrtx_amd/torch; import torch (or tensorrtx_amd) first" % ", ".join(rts))
        L.trtx_status_string.restype = ctypes.c_char_p
        L.trtx_status_string.argtypes = [ctypes.c_int32]
        L.trtx_yolo_decode_workspace.restype = ctypes.c_size_t
        _LIB = L
    return _LIB


def check(status, what):
    if status != 0:
        raise TrtxError(status, what)


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


DET_FLOATS = 90  # yolov8/include/types.h:4-12


def yolo_decode(inputs, classes, net_h, net_w, strides, max_out=1000, out=None):
    """YoloLayerPlugin::enqueue replacement (yolov8/plugin/yololayer.cu:167-316).
    inputs: list of CUDA fp32 tensors [B, 4+classes, gh*gw]. Returns [B, 1+max_out*90] fp32."""
    import torch
    L = lib()
    B = inputs[0].shape[0]
    n = len(inputs)
    ins = [x.contiguous() for x in inputs]
    for x in ins:
        assert x.is_cuda and x.dtype == torch.float32
    arr = (ctypes.c_void_p * n)(*[x.data_ptr() for x in ins])
    st = (ctypes.c_int * n)(*strides)
    ws_bytes = L.trtx_yolo_decode_workspace(B, net_h, net_w, st, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=ins[0].device)
    if out is None:
        out = torch.empty((B, 1 + max_out * DET_FLOATS), dtype=torch.float32, device=ins[0].device)
    check(L.trtx_yolo_decode(arr, n, B, classes, net_h, net_w, st, max_out, _p(out), _p(ws),
                             ctypes.c_size_t(ws_bytes), _stream()), "trtx_yolo_decode")
    return out


def yolo_nms(decode_out, max_out=1000, conf_thresh=0.5, nms_thresh=0.45, with_dets=True):
    """batch_nms replacement (yolov8/src/postprocess.cpp:71-129). Returns keep_idx, keep_cnt, keep_det."""
    import torch
    L = lib()
    B = decode_out.shape[0]
    dev = decode_out.device
    keep_idx = torch.full((B, max_out), -1, dtype=torch.int32, device=dev)
    keep_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    keep_det = torch.zeros((B, max_out, 6), dtype=torch.float32, device=dev) if with_dets else None
    L.trtx_yolo_nms_workspace.restype = ctypes.c_size_t
    ws_bytes = L.trtx_yolo_nms_workspace(B)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    check(L.trtx_yolo_nms(_p(decode_out), B, max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh),
                          _p(keep_idx), _p(keep_cnt), _p(keep_det), _p(ws), ctypes.c_size_t(ws_bytes), _stream()),
          "trtx_yolo_nms")
    return keep_idx, keep_cnt, keep_det


def yolo_decode_ex(inputs, classes, net_h, net_w, strides, max_out=1000, nk=17, kpt_conf=0.0, seg=False, pose=False, obb=False):
    """YoloLayerPlugin::enqueue with the seg / pose / obb branches (yolov8/plugin/yololayer.cu:178-279)."""
    import torch
    L = lib()
    B, n = inputs[0].shape[0], len(inputs)
    ins = [x.contiguous() for x in inputs]
    arr = (ctypes.c_void_p * n)(*[x.data_ptr() for x in ins])
    st = (ctypes.c_int * n)(*strides)
    ws_bytes = L.trtx_yolo_decode_workspace(B, net_h, net_w, st, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=ins[0].device)
    out = torch.zeros((B, 1 + max_out * DET_FLOATS), dtype=torch.float32, device=ins[0].device)
    check(L.trtx_yolo_decode_ex(arr, n, B, classes, net_h, net_w, st, max_out, nk, ctypes.c_float(kpt_conf), int(seg), int(pose), int(obb),
                                _p(out), _p(ws), ctypes.c_size_t(ws_bytes), _stream()), "trtx_yolo_decode_ex")
    return out


def yolo_nms_obb(decode_out, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """nms_obb replacement (yolov8/src/postprocess.cpp:303-393).  keep_det: [B, max_out, 7] = cx, cy, w, h, conf, cls, angle."""
    import torch
    L = lib()
    B, dev = decode_out.shape[0], decode_out.device
    keep_idx = torch.full((B, max_out), -1, dtype=torch.int32, device=dev)
    keep_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    keep_det = torch.zeros((B, max_out, 7), dtype=torch.float32, device=dev)
    L.trtx_yolo_nms_workspace.restype = ctypes.c_size_t
    ws_bytes = L.trtx_yolo_nms_workspace(B)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    check(L.trtx_yolo_nms_obb(_p(decode_out), B, max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh), _p(keep_idx), _p(keep_cnt),
                              _p(keep_det), _p(ws), ctypes.c_size_t(ws_bytes), _stream()), "trtx_yolo_nms_obb")
    return keep_idx, keep_cnt, keep_det


def yolo_postprocess_gpu_obb(decode_out, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """The reference's GPU mode for oriented boxes (yolov8/src/postprocess.cu: decode_kernel_obb + nms_kernel_obb): [B, 1 + max_out*8]."""
    import torch
    out = torch.empty((decode_out.shape[0], 1 + max_out * 8), dtype=torch.float32, device=decode_out.device)
    check(lib().trtx_yolo_postprocess_gpu_obb(_p(decode_out), decode_out.shape[0], max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh),
                                              _p(out), _stream()), "trtx_yolo_postprocess_gpu_obb")
    return out


DET5_FLOATS = 38  # yolov5/src/types.h:11-16


def yolov5_decode(inputs, classes, net_h, net_w, grids, anchors, max_out=1000, is_seg=False):
    """Anchor-based YoloLayerPlugin::enqueue replacement (yolov5/plugin/yololayer.cu:161-233).
    inputs: CUDA fp32 [B, 3*(5+classes(+32)), gh*gw] per level; grids: [(gw, gh)]; anchors: [n_levels][6].  -> [B, 1+max_out*38]"""
    import numpy as np
    import torch
    L = lib()
    n = len(inputs)
    ins = [x.contiguous() for x in inputs]
    B, dev = ins[0].shape[0], ins[0].device
    arr = (ctypes.c_void_p * n)(*[x.data_ptr() for x in ins])
    gw = (ctypes.c_int * n)(*[g[0] for g in grids])
    gh = (ctypes.c_int * n)(*[g[1] for g in grids])
    an = np.ascontiguousarray(anchors, dtype=np.float32).reshape(n, 6)
    L.trtx_yolov5_decode_workspace.restype = ctypes.c_size_t
    ws_bytes = L.trtx_yolov5_decode_workspace(B, gw, gh, n)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
    out = torch.zeros((B, 1 + max_out * DET5_FLOATS), dtype=torch.float32, device=dev)
    check(L.trtx_yolov5_decode(arr, n, B, classes, net_h, net_w, gw, gh, an.ctypes.data_as(ctypes.c_void_p), max_out, 1 if is_seg else 0,
                               _p(out), _p(ws), ctypes.c_size_t(ws_bytes), _stream()), "trtx_yolov5_decode")
    return out


def yolov5_nms(decode_out, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """yolov5 batch_nms replacement (yolov5/src/postprocess.cpp:30-80). Returns keep_idx, keep_cnt, keep_det [B, max_out, 6]."""
    import torch
    L = lib()
    B, dev = decode_out.shape[0], decode_out.device
    keep_idx = torch.full((B, max_out), -1, dtype=torch.int32, device=dev)
    keep_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    keep_det = torch.zeros((B, max_out, 6), dtype=torch.float32, device=dev)
    L.trtx_yolo_nms_workspace.restype = ctypes.c_size_t
    ws_bytes = L.trtx_yolo_nms_workspace(B)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    check(L.trtx_yolov5_nms(_p(decode_out), B, max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh), _p(keep_idx), _p(keep_cnt),
                            _p(keep_det), _p(ws), ctypes.c_size_t(ws_bytes), _stream()), "trtx_yolov5_nms")
    return keep_idx, keep_cnt, keep_det


def yolo_postprocess_gpu(decode_out, max_out=1000, conf_thresh=0.5, nms_thresh=0.45):
    """The reference's GPU post-processing mode "g" (yolov8/src/postprocess.cu:42-111): [B, 1 + max_out*7]."""
    import torch
    B = decode_out.shape[0]
    out = torch.empty((B, 1 + max_out * 7), dtype=torch.float32, device=decode_out.device)
    check(lib().trtx_yolo_postprocess_gpu(_p(decode_out), B, max_out, ctypes.c_float(conf_thresh), ctypes.c_float(nms_thresh),
                                          _p(out), _stream()), "trtx_yolo_postprocess_gpu")
    return out


ACT = {"none": 0, "relu": 1, "sigmoid": 2, "silu": 3, "leaky": 4, "tanh": 5}


def pack_conv_weights_f16(w_kcrs, cin_pad=None, ch_scale=None):
    """Host: KCRS fp32 numpy -> (packed uint16 [Cout_pad, Kpad], cout_pad, kpad, bn)."""
    import numpy as np
    L = lib()
    w = np.ascontiguousarray(w_kcrs, dtype=np.float32)
    cout, cin, kh, kw = w.shape
    cin_pad = cin_pad or (cin + 7) // 8 * 8
    cp, kp, bn = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    check(L.trtx_conv_packed_dims(cout, cin_pad, kh, kw, ctypes.byref(cp), ctypes.byref(kp), ctypes.byref(bn)),
          "trtx_conv_packed_dims")
    packed = np.zeros((cp.value, kp.value), dtype=np.uint16)
    sc = None
    if ch_scale is not None:
        sc = np.ascontiguousarray(ch_scale, dtype=np.float32)
    check(L.trtx_conv_pack_weights_f16(w.ctypes.data_as(ctypes.c_void_p), cout, cin, kh, kw, cin_pad,
                                       sc.ctypes.data_as(ctypes.c_void_p) if sc is not None else None,
                                       packed.ctypes.data_as(ctypes.c_void_p)), "trtx_conv_pack_weights_f16")
    return packed, cp.value, kp.value, bn.value


def conv2d_nhwc_f16(x, wpacked, bias, cout, kh, kw, stride, pad, act1="none", residual=None, act2="none",
                    out=None, out_ld=None):
    """Single fused conv launch on NHWC fp16 tensors (x: [N,H,W,Cin] CUDA half)."""
    import torch
    L = lib()
    N, H, W, Cin = x.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, cout), dtype=torch.float16, device=x.device)
    ld_out = out_ld or out.shape[-1]
    check(L.trtx_op_conv2d_nhwc_f16(_p(x), N, H, W, Cin, x.stride(2), _p(wpacked), _p(bias), _p(out), cout, ld_out,
                                    kh, kw, stride, stride, pad, pad, ACT[act1], _p(residual),
                                    residual.stride(2) if residual is not None else 0, ACT[act2], _stream()),
          "trtx_op_conv2d_nhwc_f16")
    return out


# ---------------------------------------------------------------------------------------------------- fp32 engines: conv on the fp32 MFMA (tests / tools)
def pack_conv_weights_f32(w_kcrs, cin_pad=None, ch_scale=None):
    """Host: KCRS fp32 numpy -> (packed float32 [Cout_pad, Kpad], cout_pad, kpad, cink) for conv2d_nhwc_f32."""
    import numpy as np
    L = lib()
    w = np.ascontiguousarray(w_kcrs, dtype=np.float32)
    cout, cin, kh, kw = w.shape
    cin_pad = cin_pad or (cin + 3) // 4 * 4
    cp, kp, ck = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    check(L.trtx_conv_packed_dims_f32(cout, cin_pad, kh, kw, ctypes.byref(cp), ctypes.byref(kp), ctypes.byref(ck)), "trtx_conv_packed_dims_f32")
    packed = np.zeros((cp.value, kp.value), dtype=np.float32)
    sc = np.ascontiguousarray(ch_scale, dtype=np.float32) if ch_scale is not None else None
    check(L.trtx_conv_pack_weights_f32(w.ctypes.data_as(ctypes.c_void_p), cout, cin, kh, kw, cin_pad,
                                       sc.ctypes.data_as(ctypes.c_void_p) if sc is not None else None,
                                       packed.ctypes.data_as(ctypes.c_void_p)), "trtx_conv_pack_weights_f32")
    return packed, cp.value, kp.value, ck.value


def conv2d_nhwc_f32(x, wpacked, bias, cout, kh, kw, stride, pad, act1="none", residual=None, act2="none", out=None, out_ld=None, tile=None):
    """Single fused conv launch on NHWC fp32 tensors (x: [N,H,W,Cin] CUDA float32, Cin % 4 == 0); tile = (bn, bm, operand path, channels per k-step) or None.
    bias: cout_pad floats (cout rounded up to 16; pack_conv_weights_f32 returns cout_pad) - the kernel reads it in 16-byte pieces up to the padded width."""
    import torch
    L = lib()
    N, H, W, Cin = x.shape
    if bias is not None and bias.numel() < (cout + 15) // 16 * 16:
        raise ValueError(f"bias holds {bias.numel()} floats, the fp32 tile reads cout_pad = {(cout + 15) // 16 * 16}")
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    ld_out = out_ld or out.shape[-1]
    t2 = (ctypes.c_int32 * 4)(*tile) if tile is not None else None
    check(L.trtx_op_conv2d_nhwc_f32(_p(x), N, H, W, Cin, x.stride(2), _p(wpacked), _p(bias), _p(out), cout, ld_out, kh, kw, stride, stride, pad, pad,
                                    ACT[act1], _p(residual), residual.stride(2) if residual is not None else 0, ACT[act2], t2, _stream()),
          "trtx_op_conv2d_nhwc_f32")
    return out


def conv2d_tactics_f32(N, H, W, Cin, Cout, k, stride, pad, residual=False, ld_in=None, ld_out=None, ld_res=None, max_out=32):
    """The launch configurations (bn, bm, operand path, channels per k-step) of one fp32 conv layer (host only); [0] is the launcher's own choice."""
    arr = (ctypes.c_int32 * (4 * max_out))()
    n = lib().trtx_op_conv2d_tactics_f32(N, H, W, Cin, ld_in or Cin, Cout, ld_out or Cout, k, k, stride, stride, pad, pad, 1 if residual else 0,
                                         (ld_res or Cout) if residual else 0, arr, max_out)
    return [(arr[4 * i], arr[4 * i + 1], arr[4 * i + 2], arr[4 * i + 3]) for i in range(n)]


# ---------------------------------------------------------------------------------------------------- test support
def poison_lds(sync=True):
    """Test support: fill every CU's LDS with fp16 NaN patterns (LDS is not cleared between kernels).  sync=False: only enqueued, on the
    current stream - the poisoning workgroups then run BESIDE whatever the other streams have in flight."""
    import torch
    w = torch.zeros(4, dtype=torch.int32, device="cuda")
    check(lib().trtx_op_poison_lds(_p(w), _stream()), "trtx_op_poison_lds")
    if sync:
        torch.cuda.synchronize()


def conv2d_tactics(N, H, W, Cin, Cout, k, stride, pad, residual=False, ld_in=None, ld_out=None, ld_res=None, max_out=32):
    """The exchangeable launch configurations of one conv layer (host only): list of (bn, bk, bm, wsk, ws, r3); [0] is the default."""
    arr = (ctypes.c_int32 * (6 * max_out))()
    n = lib().trtx_op_conv2d_tactics(N, H, W, Cin, ld_in or Cin, Cout, ld_out or Cout, k, k, stride, stride, pad, pad, 1 if residual else 0,
                                     (ld_res or Cout) if residual else 0, arr, max_out)
    return [tuple(arr[6 * i + j] for j in range(6)) for i in range(n)]


def conv_force_tactic(tactic=None):
    """Pin the launch configuration used by the following conv2d_nhwc_f16 calls of this process (None: back to the default)."""
    if tactic is None:
        check(lib().trtx_op_conv_force_tactic(None), "trtx_op_conv_force_tactic")
    else:
        check(lib().trtx_op_conv_force_tactic((ctypes.c_int32 * 6)(*tactic)), "trtx_op_conv_force_tactic")


# ---------------------------------------------------------------------------------------------------- kINT8 conv (tests / tools)
def pack_conv_weights_i8(w_kcrs, ch_scale=None):
    """Host: KCRS fp32 numpy -> (packed int8 [Cout_pad, Kpad], wscale [Cout_pad]) with per-output-channel symmetric scales."""
    import numpy as np
    L = lib()
    w = np.ascontiguousarray(w_kcrs, dtype=np.float32)
    cout, cin, kh, kw = w.shape
    cp, kp = ctypes.c_int32(), ctypes.c_int32()
    check(L.trtx_conv_pack_weights_i8(w.ctypes.data_as(ctypes.c_void_p), cout, cin, kh, kw, None, None, None, ctypes.byref(cp), ctypes.byref(kp)),
          "trtx_conv_pack_weights_i8 (dims)")
    packed = np.zeros((cp.value, kp.value), dtype=np.int8)
    wscale = np.zeros((cp.value,), dtype=np.float32)
    sc = None if ch_scale is None else np.ascontiguousarray(ch_scale, dtype=np.float32)
    check(L.trtx_conv_pack_weights_i8(w.ctypes.data_as(ctypes.c_void_p), cout, cin, kh, kw,
                                      sc.ctypes.data_as(ctypes.c_void_p) if sc is not None else None, packed.ctypes.data_as(ctypes.c_void_p),
                                      wscale.ctypes.data_as(ctypes.c_void_p), ctypes.byref(cp), ctypes.byref(kp)), "trtx_conv_pack_weights_i8")
    return packed, wscale


def conv2d_nhwc_i8(x_i8, wpacked, cscale, bias, cout, kh, kw, stride, pad, act1="none", out_scale=None, residual=None, res_scale=1.0,
                   act2="none"):
    """int8 MFMA conv: x_i8 CUDA int8 [N,H,W,Cin]; returns int8 (out_scale given: quantised with 1/out_scale) or fp16 NHWC."""
    import torch
    N, H, W, Cin = x_i8.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = torch.empty((N, Ho, Wo, cout), dtype=torch.int8 if out_scale else torch.float16, device=x_i8.device)
    res_i8 = residual is not None and residual.dtype == torch.int8
    check(lib().trtx_op_conv2d_nhwc_i8(_p(x_i8), N, H, W, Cin, x_i8.stride(2), _p(wpacked), _p(cscale), _p(bias), _p(out), 1 if out_scale else 0,
                                       ctypes.c_float(1.0 / out_scale if out_scale else 0.0), cout, cout, kh, kw, stride, stride, pad, pad, ACT[act1],
                                       _p(residual), 1 if res_i8 else 0, ctypes.c_float(res_scale), residual.stride(2) if residual is not None else 0,
                                       ACT[act2], _stream()), "trtx_op_conv2d_nhwc_i8")
    return out
