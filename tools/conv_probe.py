"""GPU probe: flat implicit-GEMM vs LDS-patch convolution on the YOLOv8n layer shapes (batch 32):
correctness against torch conv2d (small batch) and per-shape timing.  Writes gpurun_out/conv_probe.json."""
import ctypes, json, os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi
from tensorrtx_amd.capi import _p, _stream, check, ACT

dev = torch.device("cuda:0")
L = capi.lib()

def patch_pack(w, N, H, W, stride, pad):
    cout, cin, kh, kw = w.shape
    geom = (ctypes.c_int32 * 7)(); cp = ctypes.c_int32(); kp = ctypes.c_int32()
    st = L.trtx_conv_patch_plan(N, H, W, cin, cout, kh, kw, stride, pad, geom, ctypes.byref(cp), ctypes.byref(kp))
    if st != 0:
        return None, None
    packed = np.zeros((cp.value, kp.value), dtype=np.uint16)
    check(L.trtx_conv_pack_weights_patch_f16(w.ctypes.data_as(ctypes.c_void_p), cout, cin, kh, kw, cin, geom[4], None,
                                             packed.ctypes.data_as(ctypes.c_void_p)), "pack_patch")
    return packed, list(geom)

def run_patch(x, wp, bias, cout, k, s, p, act, out, res=None):
    N, H, W, Cin = x.shape
    check(L.trtx_op_conv2d_nhwc_f16_patch(_p(x), N, H, W, Cin, x.stride(2), _p(wp), _p(bias), _p(out), cout, out.shape[-1], k, k, s, p,
                                          ACT[act], _p(res), res.stride(2) if res is not None else 0, 0, _stream()), "patch")

SHAPES = [(16, 32, 3, 2, 320), (32, 32, 1, 1, 160), (16, 16, 3, 1, 160), (48, 32, 1, 1, 160), (32, 64, 3, 2, 160), (64, 64, 1, 1, 80),
          (32, 32, 3, 1, 80), (128, 64, 1, 1, 80), (64, 128, 3, 2, 80), (128, 128, 1, 1, 40), (64, 64, 3, 1, 40), (256, 128, 1, 1, 40),
          (128, 256, 3, 2, 40), (256, 256, 1, 1, 20), (128, 128, 3, 1, 20), (384, 256, 1, 1, 20), (512, 256, 1, 1, 20), (384, 128, 1, 1, 40),
          (192, 128, 1, 1, 40), (192, 64, 1, 1, 80), (96, 64, 1, 1, 80), (64, 64, 3, 2, 80), (128, 128, 3, 2, 40),
          (64, 64, 3, 1, 80), (64, 80, 3, 1, 80), (80, 80, 3, 1, 80), (80, 80, 1, 1, 80), (128, 64, 3, 1, 40), (128, 80, 3, 1, 40),
          (80, 80, 3, 1, 40), (256, 64, 3, 1, 20), (256, 80, 3, 1, 20), (64, 64, 3, 1, 20), (80, 80, 3, 1, 20)]
rows = []
for (cin, cout, k, s, hin) in SHAPES:
    p = k // 2
    rng = np.random.default_rng(1)
    w = rng.normal(0, (2.0 / (cin * k * k)) ** 0.5, size=(cout, cin, k, k)).astype(np.float32)
    bias_np = rng.normal(0, 0.1, size=(cout,)).astype(np.float32)
    # correctness at batch 2
    xs = torch.randn(2, hin, hin, cin, device=dev).half()
    ref = F.silu(F.conv2d(xs.float().permute(0, 3, 1, 2), torch.from_numpy(w).to(dev).half().float(), torch.from_numpy(bias_np).to(dev), stride=s, padding=p)).permute(0, 2, 3, 1)
    ho = ref.shape[1]
    pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
    wp_i = torch.from_numpy(pk.view(np.int16)).to(dev)
    bias = torch.zeros(cp, device=dev); bias[:cout] = torch.from_numpy(bias_np).to(dev)
    y_i = capi.conv2d_nhwc_f16(xs, wp_i, bias, cout, k, k, s, p, "silu")
    err_i = (y_i.float() - ref).abs().max().item()
    pp, geom = patch_pack(w, 2, hin, hin, s, p)
    err_p = None
    if pp is not None:
        wp_p = torch.from_numpy(pp.view(np.int16)).to(dev)
        y_p = torch.empty_like(y_i)
        run_patch(xs, wp_p, bias, cout, k, s, p, "silu", y_p)
        torch.cuda.synchronize()
        err_p = (y_p.float() - ref).abs().max().item()
    # timing at batch 32
    x = torch.randn(32, hin, hin, cin, device=dev).half()
    y = torch.empty(32, ho, ho, cout, device=dev, dtype=torch.float16)
    def timeit(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    t_i = timeit(lambda: capi.conv2d_nhwc_f16(x, wp_i, bias, cout, k, k, s, p, "silu", out=y))
    t_p = None
    if pp is not None:
        pp32, geom = patch_pack(w, 32, hin, hin, s, p)
        wp_p = torch.from_numpy(pp32.view(np.int16)).to(dev)
        t_p = timeit(lambda: run_patch(x, wp_p, bias, cout, k, s, p, "silu", y))
    flop = 2.0 * 32 * ho * ho * cout * cin * k * k
    byts = 2.0 * (x.numel() + y.numel())
    row = dict(cin=cin, cout=cout, k=k, s=s, hin=hin, err_igemm=err_i, err_patch=err_p, us_igemm=t_i, us_patch=t_p, geom=geom,
               tflops_igemm=flop / t_i / 1e6, tflops_patch=(flop / t_p / 1e6 if t_p else None), hbm_floor_us=byts / 5e6)
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/conv_probe.json", "w"), indent=0)
print("sum igemm us", sum(r["us_igemm"] for r in rows), "sum best us", sum(min(r["us_igemm"], r["us_patch"] or 1e9) for r in rows))
