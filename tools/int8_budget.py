"""Per-tensor INT8 budget of the YOLOv8n engine under the reference's entropy calibrator - CPU only.

The fused-plan emulator of tools/fp16_budget.py with, at every activation site an int8 engine keeps in int8, the engine's requantisation:
q = clamp(round(x / s), -127, 127) * s with s = threshold / 127, the threshold from oracle/quant.py::entropy_threshold (the KL search of
IInt8EntropyCalibrator2, yolov8/src/calibrator.cpp feeds it) or the largest |x| seen (IInt8MinMaxCalibrator), both from 2048-bin |x|
histograms of the SAME calibration images the GPU test uses (tests/test_gpu_int8.py: synth.images(4, 640, 640, seed=60..63)); int8 weights
per output channel (max |w| / 127, nothing clipped).  Judged like the GPU test: the fp32 oracle's candidates with conf > 0.25 looked up by
class and IoU > 0.5 among the emulated engine's (seed-1 images the calibration did not see).

Questions answered, with numbers: how much of the loss is CLIPPING and how much is the 8-bit grid; which tensors carry it ("alone": only
this tensor int8 with its entropy threshold, the rest fp16; "relieved": every tensor int8 / entropy except this one on min-max); how many
tensors would have to leave entropy calibration for >= 90 % of the candidates to be found again.

    python tools/int8_budget.py [--out profiles/r04_int8_budget.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import fp16_budget as fb  # noqa: E402
from oracle import quant  # noqa: E402
from oracle import wts as owts  # noqa: E402
from oracle import yolo_post as yp  # noqa: E402
from tensorrtx_amd import synth  # noqa: E402

NB = 2048


def det_stats(dec, dec_ref, conf_floor=0.25):
    """tests/test_gpu_int8.py::_yolo_detection_stats, same rule"""
    total = hit50 = 0
    ious = []
    for b in range(dec_ref.shape[0]):
        nr, ng = int(dec_ref[b, 0]), int(dec[b, 0])
        R = dec_ref[b, 1:1 + nr * 90].reshape(nr, 90)
        G = dec[b, 1:1 + ng * 90].reshape(ng, 90)
        for r in R[R[:, 4] > conf_floor]:
            total += 1
            cand = G[G[:, 5] == r[5]]
            if not len(cand):
                continue
            ix = np.maximum(0, np.minimum(cand[:, 2], r[2]) - np.maximum(cand[:, 0], r[0]))
            iy = np.maximum(0, np.minimum(cand[:, 3], r[3]) - np.maximum(cand[:, 1], r[1]))
            inter = ix * iy
            iou = inter / ((cand[:, 2] - cand[:, 0]) * (cand[:, 3] - cand[:, 1]) + (r[2] - r[0]) * (r[3] - r[1]) - inter + 1e-12)
            if iou.max() > 0.5:
                hit50 += 1
                ious.append(float(iou.max()))
    return hit50 / max(total, 1), (float(np.mean(ious)) if ious else 0.0), total


def int8_site(name):
    """activation sites the int8 engine holds in int8 (runtime/lower.cpp assign_int8): conv outputs read by MFMA convolutions.  Not: the
    stem's output (conv_stem has no requantising epilogue), SPPF's cv1 (the pool chain reads it), the detect convs' outputs (the fused
    head reads fp16), the fp16 value before a shortcut add."""
    if not name.startswith("a:") or name.endswith("(pre-add)"):
        return False
    n = name[2:]
    return not (n == "model.0" or n == "model.9.cv1" or (n.startswith("model.22.") and n.endswith(".2")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--size", type=int, default=640)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    from util import synth_wts
    path, _ = synth_wts("yolov8n")
    tensors = owts.load_wts(path)
    S = a.size
    lines = []

    def say(s=""):
        print(s, flush=True)
        lines.append(s)

    with torch.inference_mode():
        # ---- calibration: |x| histograms of every int8 site over the GPU test's 16 calibration images, fp16 engine arithmetic
        t0 = time.time()
        acts = {}

        def collect(name):
            def f(t):
                t = fb.r16(t)
                if int8_site(name):
                    acts.setdefault(name, []).append(t.abs().flatten())
                return t
            return f
        for k in range(4):
            fb.forward(tensors, torch.from_numpy(synth.images(4, S, S, seed=60 + k)), collect)
        thr_e, thr_m, clipped = {}, {}, {}
        for name, parts in acts.items():
            v = torch.cat(parts)
            rng = float(v.max())
            h = torch.histc(v, bins=NB, min=0.0, max=rng).numpy()
            thr_m[name] = rng
            thr_e[name] = float(quant.entropy_threshold(h, rng))
            clipped[name] = float((v > thr_e[name]).float().mean())
        del acts
        say(f"# INT8 budget of the YOLOv8n engine under entropy calibration, CPU emulation (tools/int8_budget.py); {len(thr_e)} int8 activation tensors; "
            f"calibration {time.time() - t0:.0f} s")

        x = torch.from_numpy(synth.images(4, S, S, seed=1))
        ref, sites = fb.forward(tensors, x, lambda s: False)
        dec_ref = yp.decode_c([h.numpy() for h in ref], 80, S, S, [8, 16, 32])

        def wq(w):   # per-output-channel symmetric int8 weights (conv_pack_weights_i8)
            s = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30) / 127.0
            return torch.round(w / s).clamp(-127, 127) * s

        def measure(thr_of, clip=True, grid=True, int8_weights=True):
            """thr_of(name) -> threshold of an int8 activation site, or None to leave it in fp16"""
            def on(name):
                if name.startswith("w:"):
                    n = name[2:]
                    # weights of convolutions whose INPUT is int8 (approximation: every conv but the stem and those fed by fp16-only tensors)
                    if int8_weights and n != "model.0":
                        return lambda w: wq(fb.r16(w))
                    return True
                thr = thr_of(name) if int8_site(name) else None
                if thr is None:
                    return True

                def f(t, thr=thr):
                    t = fb.r16(t)
                    s = thr / 127.0
                    q = t / s
                    if grid:
                        q = torch.round(q)
                    if clip:
                        q = q.clamp(-127, 127)
                    return fb.r16(q * s)
                return f
            out, _ = fb.forward(tensors, x, on)
            dec = yp.decode_c([o.numpy() for o in out], 80, S, S, [8, 16, 32])
            return det_stats(dec, dec_ref)

        fp16 = measure(lambda n: None, int8_weights=False)
        e_all = measure(lambda n: thr_e[n])
        m_all = measure(lambda n: thr_m[n])
        say(f"# candidates (fp32 oracle, conf > 0.25): {e_all[2]}")
        say(f"fp16 engine (emulated)                                  matched@0.5 {fp16[0]:.4f}  mean IoU {fp16[1]:.4f}")
        say(f"int8, entropy thresholds (the reference's calibrator)   matched@0.5 {e_all[0]:.4f}  mean IoU {e_all[1]:.4f}     measured on MI355X: 0.69 / 0.83")
        say(f"int8, min-max thresholds                                matched@0.5 {m_all[0]:.4f}  mean IoU {m_all[1]:.4f}     measured on MI355X: 0.98 / 0.96")
        c_only = measure(lambda n: thr_e[n], clip=True, grid=False, int8_weights=False)
        g_only = measure(lambda n: thr_e[n], clip=False, grid=True)
        say(f"entropy thresholds, CLIPPING only (no 8-bit grid)       matched@0.5 {c_only[0]:.4f}  mean IoU {c_only[1]:.4f}")
        say(f"entropy scales, 8-bit GRID only (nothing clipped)       matched@0.5 {g_only[0]:.4f}  mean IoU {g_only[1]:.4f}")
        say()
        say("## per tensor.  alone: ONLY this tensor int8 at its entropy threshold, everything else fp16.  relieved: everything int8 / entropy, this tensor on min-max")
        say(f"{'tensor':34s} {'absmax':>8s} {'entropy thr':>11s} {'thr/max':>7s} {'clipped':>9s} | {'alone matched':>13s} | {'relieved matched':>16s} {'gain':>7s}")
        rows = []
        for n in thr_e:
            al = measure(lambda s, n=n: thr_e[n] if s == n else None, int8_weights=False)
            rl = measure(lambda s, n=n: thr_m[s] if s == n else thr_e[s])
            rows.append((n, al[0], rl[0]))
            say(f"{n[2:]:34s} {thr_m[n]:8.3f} {thr_e[n]:11.3f} {thr_e[n] / thr_m[n]:7.3f} {clipped[n]:9.2e} | {al[0]:13.4f} | {rl[0]:16.4f} {rl[0] - e_all[0]:+7.4f}")
        say()
        say("## cumulative: tensors moved from entropy to min-max in the order of their 'alone' loss, until 90 % of the candidates are found again")
        order = sorted(rows, key=lambda r: r[1])
        moved = set()
        for k, (n, al, _) in enumerate(order):
            moved.add(n)
            if (k + 1) in (1, 2, 3, 5, 8, 12, 16, 24, 32, 40, len(order)) or k == len(order) - 1:
                r = measure(lambda s: thr_m[s] if s in moved else thr_e[s])
                say(f"{k + 1:3d} tensors on min-max (last added {n[2:]:28s} alone {al:.4f})   matched@0.5 {r[0]:.4f}  mean IoU {r[1]:.4f}")
                if r[0] >= 0.9 and k + 1 < len(order):
                    say(f"   -> {k + 1} of {len(order)} tensors have to leave entropy calibration for 0.90")
                    break
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
