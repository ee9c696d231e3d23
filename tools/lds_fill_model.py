"""How many operand bytes one step pushes through the LDS fill path (global -> LDS), from the lowered plan alone (CPU): the implicit-GEMM kernels
fetch an A tile per (tile, tap, channel slice) - a 3x3 layer's input crosses the path 9x per column tile - and a B tile per (row tile, k-step).
    bytes(layer) = M * Kpad * 2 * (Cout_pad / BN)  +  ceil(M / BM) * Cout_pad * Kpad * 2
against the layer's algorithmic HBM bytes.  Printed next to it: what the same layer would push with (a) row reuse (one A tile per filter ROW: the
row-reuse kernel, A / 3 for 3x3 stride 1), (b) a resident input patch (tile + halo once, nine taps read from LDS: what conv_ws does for the
small-channel layers - modelled as A = M * CinK * 2 * 1.3 per column tile), (c) 256-row tiles (B / 2).
A model of BYTES, not of time: the fill RATE depends on the path (LDS-DMA measured at ~15 B/clk per CU for a lone workgroup, profiles/r04_launch_anatomy.txt;
the register path is faster).  python tools/lds_fill_model.py [model batch h w]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tensorrtx_amd import engine  # noqa: E402
from util import synth_wts  # noqa: E402


def pick_bn(cout_pad16):
    for bn in (128, 80, 64, 32, 16):
        if cout_pad16 % bn == 0:
            return bn
    return 16


def main(model="yolov8n", B=32, H=640, W=640):
    path, _ = synth_wts(model)
    d = engine.describe_plan(engine.build_plan(model, path, batch=B, h=H, w=W, fp16=1, aux_streams=0), lowered=True)
    rows = []

    def one(o):
        if o.get("stem") or not o.get("igemm"):
            return
        cin, cout = o["cin"], o["cout"]
        kh, kw = o["k"]
        Ho, Wo = o["hw_out"]
        M = B * o.get("nmul", 1) * Ho * Wo
        cink = 16 if cin <= 16 else (cin + 31) // 32 * 32
        K = kh * kw * cink
        bn = pick_bn((cout + 15) // 16 * 16)
        coutp = (cout + bn - 1) // bn * bn
        ct = coutp / bn
        a = M * K * 2 * ct
        b = math.ceil(M / 128) * coutp * K * 2
        s1 = o["stride"] == [1, 1]
        ws = kh == 3 and s1 and cin <= 32 and cout <= 32 and M // 128 >= 1024   # today on conv_ws: resident patch, weights in registers (the six 16 / 32-channel 3x3 layers of YOLOv8n)
        if ws:
            a, b = M * cink * 2 * 1.3 * ct, 0.0
        a_row = a / 3 if (kh == 3 and kw == 3 and s1 and not ws) else a
        a_patch = M * cink * 2 * 1.3 * ct if (kh == 3 and kw == 3 and s1) else a
        # the resident-patch kernel AS WRITTEN (conv_igemm.hip, experimental build): 64 / 80 / 128-wide column tiles, <= 128 or exactly 256 input channels;
        # 16-row tiles (B per 256 pixels) up to 64 channels on maps of 16+ rows, else 8-row tiles; patch = (TH + 2) x 18 pixels per TH x 16 outputs
        elig = kh == 3 and kw == 3 and s1 and not ws and bn in (64, 80, 128) and cink % 32 == 0 and (cink <= 128 or cink == 256)
        if elig:
            th = 16 if (cink <= 64 and Ho >= 16) else 8
            a_kern = M * cink * 2 * ((th + 2) * 18 / (th * 16.0)) * ct
            b_kern = math.ceil(M / (th * 16)) * coutp * K * 2
        else:
            a_kern, b_kern = a, b
        rows.append(dict(name=o["name"][-24:] + (" [ws]" if ws else (" [p]" if elig else "")), a_kern=a_kern, b_kern=b_kern, cin=cin, cout=cout, k=kh, hw=Ho, bn=bn, a=a, b=b, a_row=a_row, a_patch=a_patch, hbm=o["bytes"] * B,
                         flop=o["flops"] * B))

    for o in d["ops"]:
        if o["kind"] == "conv":
            one(o)
        elif o["kind"] == "conv_group":
            for m in o["members"]:
                one(m)
    T = lambda k: sum(r[k] for r in rows)  # noqa: E731
    print(f"# {model} b{B} {H}x{W}: {len(rows)} implicit-GEMM convolution layers (static default tiles: 128 rows, the widest column tile that divides Cout), "
          f"{T('flop') / 1e9:.1f} GFLOP per step")
    print(f"algorithmic HBM bytes of these layers          {T('hbm') / 1e9:6.2f} GB")
    print(f"through the LDS fill path today: A {T('a') / 1e9:.2f} + B {T('b') / 1e9:.2f} = {(T('a') + T('b')) / 1e9:6.2f} GB  ({(T('a') + T('b')) / T('hbm'):.1f}x the HBM bytes)")
    for label, a, b in (("row reuse on the 3x3 stride-1 layers", T("a_row"), T("b")), ("resident patch on the 3x3 stride-1 layers", T("a_patch"), T("b")),
                        ("resident patch + 256-row tiles", T("a_patch"), T("b") / 2)):
        print(f"  with {label:44s} A {a / 1e9:.2f} + B {b / 1e9:.2f} = {(a + b) / 1e9:6.2f} GB")
    print(f"  with the resident-patch kernel as written, on the layers it takes ([p] below): A {T('a_kern') / 1e9:.2f} + B {T('b_kern') / 1e9:.2f} = {(T('a_kern') + T('b_kern')) / 1e9:6.2f} GB")
    for rate in (15, 30, 60):
        print(f"fill time at {rate:2d} B/clk per CU x 256 CUs x 2.4 GHz: today {(T('a') + T('b')) / (rate * 256 * 2.4e9) * 1e3:.3f} ms")
    print("\nlargest layers (MB through the fill path: A, B | algorithmic HBM MB | GFLOP)")
    for r in sorted(rows, key=lambda r: -(r["a"] + r["b"]))[:14]:
        print(f"  {r['name']:24s} {r['cin']:4d} -> {r['cout']:4d} k{r['k']} @{r['hw']:3d} bn {r['bn']:3d}   A {r['a'] / 1e6:6.0f}  B {r['b'] / 1e6:6.0f} | {r['hbm'] / 1e6:6.1f} | {r['flop'] / 1e9:5.1f}")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(*([a[0], int(a[1]), int(a[2]), int(a[3])] if len(a) >= 4 else []))
