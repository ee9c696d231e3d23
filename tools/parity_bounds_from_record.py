"""Suggest tests/parity.py bounds from a record of GPU runs: for every bounded metric the worst recorded value and a bound at 1.3x the
worst (shortfall from 1 for "min" metrics), rounded to 3 significant digits.  python tools/parity_bounds_from_record.py profiles/r03_parity_metrics.jsonl"""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import parity  # noqa: E402


def sig(x, n=3, up=True):
    if x == 0:
        return 0.0
    e = math.floor(math.log10(abs(x))) - (n - 1)
    f = 10.0 ** e
    return (math.ceil(x / f) if up else math.floor(x / f)) * f


rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
for key, metrics in parity.BOUNDS.items():
    mine = [r for r in rows if r["test"] == key[0] and r.get("case") == key[1]]
    if not mine:
        print(key, "NO RECORD")
        continue
    out = {}
    for m, spec in metrics.items():
        vals = [r[m] for r in mine]
        if spec[0] == "max":
            worst = max(vals)
            out[m] = ("max", float(f"{sig(1.3 * worst):.6g}"), f"worst {worst:.6g} now {spec[1]}")
        else:
            worst = min(vals)
            slack = spec[2] if len(spec) > 2 else 0.0
            b = 1 - (1.3 * (1 - worst) + (slack if worst == 1.0 else 0.0))
            out[m] = ("min", float(f"{1 - sig(1 - b, 3, up=True):.6g}") if b < 1 else 1.0, f"worst {worst:.6g} now {spec[1]}")
    print(key, len(mine), "runs")
    for m, v in out.items():
        print("    ", m, v)
