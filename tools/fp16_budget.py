"""Per-tensor fp16 error budget of the YOLOv8n fp16 engine - CPU only, no GPU minutes.

An EMULATOR of the engine's arithmetic on the CPU: the network of oracle/models_torch.py::yolov8_det evaluated the way the lowered plan
evaluates it - BatchNorm folded into the convolution weights, the folded weights rounded to fp16, fp32 accumulation, ONE fp16 rounding at
the output of every fused launch (after bias / SiLU / shortcut add), the input image rounded to fp16 by the stem - with every one of
those roundings individually switchable.  Rounding sites:

    in                the image (the stem converts fp32 pixels to fp16 MFMA operands)
    w:<layer>         the folded weights of one convolution
    a:<layer>         the output tensor of one fused launch (the only place an activation is rounded)

For each site the tool measures, against the all-fp32 evaluation of the same images,
    cls   max |class logit error| over all cells          (north_star: 1e-4)
    dfl   max |DFL distance error| in cells
    iou   min IoU of the decoded candidates (conf >= 0.1 + margin, matched by cell) and the matched fraction   (north_star: 1 - 1e-3)
with ONLY that site rounding ("alone") and with every site BUT that one rounding ("all-but").  Sites are also grouped (backbone, neck,
box arms, class arms, last 1x1) and the candidate mitigation sets are evaluated: which set of tensors / weights would have to stay fp32
for min IoU >= 0.999.

    python tools/fp16_budget.py [--batch 4] [--seed 1] [--out profiles/r04_fp16_budget.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import models_torch as mt  # noqa: E402
from oracle import wts as owts  # noqa: E402
from oracle import yolo_post as yp  # noqa: E402
from tensorrtx_amd import synth  # noqa: E402


def r16(t):
    return t.half().float()


class Emu:
    """The fused-plan evaluation with switchable rounding sites.  `on(site)` decides whether a site rounds."""

    def __init__(self, tensors, on):
        self.p = mt.Params(tensors)
        self.on = on
        self.sites = []

    def site(self, name, t):
        if name not in self.sites:
            self.sites.append(name)
        r = self.on(name)
        if callable(r):      # tools/int8_budget.py: a site may carry its own transform (quantise / clip) instead of the fp16 rounding
            return r(t)
        return r16(t) if r else t

    def conv(self, x, name, cout, k, s, pad, bn=True, act=True, gain=2.0, bias_init=None, res=None):
        p = self.p
        if bn:
            w = p.conv_w(name + ".conv.weight", cout, x.shape[1], k)
            g, b, m, v = p.bn(name + ".bn", cout)
            scale = g / torch.sqrt(v + 1e-3)           # block.cpp:79-96 (float sqrt), folded as runtime/lower.cpp does
            w = w * scale[:, None, None, None]
            bias = b - m * scale
        else:
            w = p.conv_w(name + ".weight", cout, x.shape[1], k, gain=gain)
            bias = p.vec(name + ".bias", cout, bias_init)
        w = self.site("w:" + name, w)
        y = F.conv2d(x, w, bias, stride=s, padding=pad)
        if act:
            y = y * torch.sigmoid(y)
        if res is not None:
            # the engine rounds the activation, then adds the (already fp16) shortcut in fp32 and rounds once more
            y = self.site("a:" + name + "(pre-add)", y) + res
        return self.site("a:" + name, y)


def forward(tensors, x, on, num_class=80):
    e = Emu(tensors, on)
    W = lambda v: mt._get_width(v, 0.25, 1024)  # noqa: E731
    D = lambda v: mt._get_depth(v, 0.33)  # noqa: E731
    x = e.site("in", x)

    def cbs(x, name, cout, k, s, pad, res=None):
        return e.conv(x, name, cout, k, s, pad, res=res)

    def c2f(x, c2, n, shortcut, name):
        c_ = c2 // 2
        y = cbs(x, name + ".cv1", 2 * c_, 1, 1, 0)
        parts = [y[:, :c_], y[:, c_:]]
        cur = parts[1]
        for i in range(n):
            t = cbs(cur, f"{name}.m.{i}.cv1", c_, 3, 1, 1)
            cur = cbs(t, f"{name}.m.{i}.cv2", c_, 3, 1, 1, res=cur if shortcut else None)
            parts.append(cur)
        return cbs(torch.cat(parts, 1), name + ".cv2", c2, 1, 1, 0)

    def sppf(x, c1, c2, name):
        y = cbs(x, name + ".cv1", c1 // 2, 1, 1, 0)
        ps = [y]
        for _ in range(3):
            ps.append(F.max_pool2d(ps[-1], 5, 1, 2))
        return cbs(torch.cat(ps, 1), name + ".cv2", c2, 1, 1, 0)

    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")  # noqa: E731
    p1 = cbs(x, "model.0", W(64), 3, 2, 1)
    p2 = cbs(p1, "model.1", W(128), 3, 2, 1)
    c2 = c2f(p2, W(128), D(3), True, "model.2")
    p3 = cbs(c2, "model.3", W(256), 3, 2, 1)
    c4 = c2f(p3, W(256), D(6), True, "model.4")
    p4 = cbs(c4, "model.5", W(512), 3, 2, 1)
    c6 = c2f(p4, W(512), D(6), True, "model.6")
    p5 = cbs(c6, "model.7", W(1024), 3, 2, 1)
    c8 = c2f(p5, W(1024), D(3), True, "model.8")
    c9 = sppf(c8, W(1024), W(1024), "model.9")
    c12 = c2f(torch.cat([up(c9), c6], 1), W(512), D(3), False, "model.12")
    c15 = c2f(torch.cat([up(c12), c4], 1), W(256), D(3), False, "model.15")
    c16 = cbs(c15, "model.16", W(256), 3, 2, 1)
    c18 = c2f(torch.cat([c16, c12], 1), W(512), D(3), False, "model.18")
    c19 = cbs(c18, "model.19", W(512), 3, 2, 1)
    c21 = c2f(torch.cat([c19, c9], 1), W(1024), D(3), False, "model.21")
    outs = []
    for lv, feat in enumerate((c15, c18, c21)):
        s = str(lv)
        b = cbs(cbs(feat, f"model.22.cv2.{s}.0", 64, 3, 1, 1), f"model.22.cv2.{s}.1", 64, 3, 1, 1)
        box = e.conv(b, f"model.22.cv2.{s}.2", 64, 1, 1, 0, bn=False, act=False, gain=4.0, bias_init=None)
        k = cbs(cbs(feat, f"model.22.cv3.{s}.0", 80, 3, 1, 1), f"model.22.cv3.{s}.1", 80, 3, 1, 1)
        cls = e.conv(k, f"model.22.cv3.{s}.2", num_class, 1, 1, 0, bn=False, act=False, gain=80.0, bias_init=None)
        g = box.shape[2] * box.shape[3]
        t = box.reshape(-1, 4, 16, g).permute(0, 2, 1, 3)
        dfl_w = torch.arange(16.0).reshape(1, 16, 1, 1)
        t = F.conv2d(F.softmax(t, dim=1), dfl_w).reshape(-1, 4, g)
        outs.append(torch.cat([t, cls.reshape(cls.shape[0], num_class, g)], 1).contiguous())
    return outs, e.sites


def iou_stats(dec, dec_ref, conf_margin=0.02):
    """tests/test_gpu_engine.py::_match_detections, same rule."""
    st = dict(ref=0, matched=0, min_iou=1.0)
    for b in range(dec_ref.shape[0]):
        nr, ng = int(dec_ref[b, 0]), int(dec[b, 0])
        R = dec_ref[b, 1:1 + nr * 90].reshape(nr, 90)[:, :6]
        G = dec[b, 1:1 + ng * 90].reshape(ng, 90)[:, :6]
        if nr >= 1000 or ng >= 1000:
            continue
        for r in R:
            if abs(r[4] - 0.1) < conf_margin:
                continue
            st["ref"] += 1
            same = np.nonzero(G[:, 5] == r[5])[0]
            if len(same) == 0:
                continue
            c = np.abs((G[same, 0] + G[same, 2]) - (r[0] + r[2])) + np.abs((G[same, 1] + G[same, 3]) - (r[1] + r[3]))
            g = G[same[np.argmin(c)]]
            ix = max(0.0, min(r[2], g[2]) - max(r[0], g[0])) * max(0.0, min(r[3], g[3]) - max(r[1], g[1]))
            ua = (r[2] - r[0]) * (r[3] - r[1]) + (g[2] - g[0]) * (g[3] - g[1]) - ix
            iou = ix / ua if ua > 0 else 0.0
            if iou > 0.9:
                st["matched"] += 1
                st["min_iou"] = min(st["min_iou"], float(iou))
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="groups and mitigation sets only (no per-site sweep)")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    from util import synth_wts
    path, _ = synth_wts("yolov8n")
    tensors = owts.load_wts(path)
    x = torch.from_numpy(synth.images(a.batch, a.size, a.size, seed=a.seed))
    strides = [8, 16, 32]
    lines = []

    def say(s=""):
        print(s, flush=True)
        lines.append(s)

    with torch.inference_mode():
        t0 = time.time()
        ref, sites = forward(tensors, x, lambda s: False)
        dt = time.time() - t0
        # the emulator with nothing rounding must BE the oracle network
        heads, _ = mt.yolov8_det(mt.Params(tensors), x)
        fold_err = max((r - h).abs().max().item() for r, h in zip(ref, heads))
        dec_ref = yp.decode_c([h.numpy() for h in ref], 80, a.size, a.size, strides)

        def measure(on):
            out, _ = forward(tensors, x, on)
            cls = max((o[:, 4:] - r[:, 4:]).abs().max().item() for o, r in zip(out, ref))
            dfl = max((o[:, :4] - r[:, :4]).abs().max().item() for o, r in zip(out, ref))
            dec = yp.decode_c([o.numpy() for o in out], 80, a.size, a.size, strides)
            st = iou_stats(dec, dec_ref)
            return cls, dfl, st["min_iou"], st["matched"] / max(st["ref"], 1), st["ref"]

        say(f"# fp16 error budget of the YOLOv8n engine, CPU emulation of the fused plan's arithmetic (tools/fp16_budget.py)")
        say(f"# images: synth.images({a.batch}, {a.size}, {a.size}, seed={a.seed}); weights: the seeded synthetic .wts; {len(sites)} rounding sites; one forward {dt:.1f} s")
        say(f"# BN-folded fp32 evaluation vs oracle/models_torch.py: max |head diff| {fold_err:.2e} (folding itself, fp32 rounding)")
        full = measure(lambda s: True)
        say(f"# ALL sites rounding (= the fp16 engine):  cls {full[0]:.4f}  dfl {full[1]:.4f} cells  min IoU {full[2]:.5f}  matched {full[3]:.5f} of {full[4]}")
        say(f"#   measured on MI355X (profiles/r03_parity_metrics.jsonl, same images): cls 0.065-0.077, dfl 0.015-0.020, min IoU 0.9967-0.9975")
        say()

        def group_of(s):
            kind, name = s.split(":", 1) if ":" in s else ("in", s)
            name = name.replace("(pre-add)", "")
            if s == "in":
                return "input"
            idx = int(name.split(".")[1])
            if idx <= 9:
                part = "backbone(0-9)"
            elif idx <= 21:
                part = "neck(12-21)"
            elif ".cv2." in name:
                part = "box-arm-last1x1" if name.endswith(".2") else "box-arms(cv2.x.0/1)"
            else:
                part = "cls-arm-last1x1" if name.endswith(".2") else "cls-arms(cv3.x.0/1)"
            return ("W " if kind == "w" else "A ") + part

        groups = {}
        for s in sites:
            groups.setdefault(group_of(s), []).append(s)
        say("## groups: only this group rounds ('alone') / everything but this group rounds ('all-but')")
        say(f"{'group':34s} {'sites':>5s} | {'cls alone':>9s} {'dfl alone':>9s} {'minIoU alone':>12s} | {'cls all-but':>11s} {'dfl all-but':>11s} {'minIoU all-but':>14s}")
        for gname, ss in groups.items():
            sset = set(ss)
            al = measure(lambda s: s in sset)
            ab = measure(lambda s: s not in sset)
            say(f"{gname:34s} {len(ss):5d} | {al[0]:9.5f} {al[1]:9.5f} {al[2]:12.6f} | {ab[0]:11.5f} {ab[1]:11.5f} {ab[2]:14.6f}")
        say()
        say("## candidate fp32 sets: everything rounds EXCEPT the listed sites (what the engine would keep in fp32)")
        def in_groups(*names):
            return set(s for n in names for s in groups.get(n, []))
        cands = {
            "last 1x1 outputs fp32 (cv2.x.2, cv3.x.2 activations)": in_groups("A box-arm-last1x1", "A cls-arm-last1x1"),
            "box-arm last 1x1: output + weights fp32": in_groups("A box-arm-last1x1", "W box-arm-last1x1"),
            "whole box arms fp32 (cv2.*: activations + weights)": in_groups("A box-arm-last1x1", "W box-arm-last1x1", "A box-arms(cv2.x.0/1)", "W box-arms(cv2.x.0/1)"),
            "box arms + neck activations fp32": in_groups("A box-arm-last1x1", "W box-arm-last1x1", "A box-arms(cv2.x.0/1)", "W box-arms(cv2.x.0/1)", "A neck(12-21)"),
            "box arms + neck (activations + weights) fp32": in_groups("A box-arm-last1x1", "W box-arm-last1x1", "A box-arms(cv2.x.0/1)", "W box-arms(cv2.x.0/1)", "A neck(12-21)", "W neck(12-21)"),
            "all activations fp32 (weights fp16)": set(s for s in sites if s.startswith("a:") or s == "in"),
            "all weights fp32 (activations fp16)": set(s for s in sites if s.startswith("w:")),
            "backbone fp32 (activations + weights + input)": in_groups("input", "A backbone(0-9)", "W backbone(0-9)"),
        }
        say(f"{'fp32 set':58s} {'sites':>5s} | {'cls':>8s} {'dfl':>8s} {'min IoU':>9s} {'matched':>8s}")
        for cname, keep in cands.items():
            r = measure(lambda s: s not in keep)
            say(f"{cname:58s} {len(keep):5d} | {r[0]:8.5f} {r[1]:8.5f} {r[2]:9.6f} {r[3]:8.5f}")
        say()
        if not a.quick:
            say("## per site: only this site rounds / every site but this one rounds")
            say(f"{'site':44s} | {'cls alone':>9s} {'dfl alone':>9s} {'1-minIoU alone':>14s} | {'cls all-but':>11s} {'dfl all-but':>11s} {'1-minIoU all-but':>16s}")
            for s in sites:
                al = measure(lambda t: t == s)
                ab = measure(lambda t: t != s)
                say(f"{s:44s} | {al[0]:9.5f} {al[1]:9.5f} {1 - al[2]:14.2e} | {ab[0]:11.5f} {ab[1]:11.5f} {1 - ab[2]:16.2e}")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
