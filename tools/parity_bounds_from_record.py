"""tests/parity.py bounds from a record of GPU runs: for every metric of parity.SPEC the worst recorded value and a bound at 1.4x the worst
(for "min" metrics: 1.4x the shortfall from 1, plus the metric's slack), rounded outward to 3 significant digits.
    python tools/parity_bounds_from_record.py profiles/r03_parity_metrics.jsonl            # print
    python tools/parity_bounds_from_record.py --write profiles/r03_parity_metrics.jsonl    # rewrite the VALUES table of tests/parity.py"""
import json
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import parity  # noqa: E402

FACTOR = 1.4


def sig(x, n=3):
    """round |x| UP to n significant digits"""
    if x == 0:
        return 0.0
    f = 10.0 ** (math.floor(math.log10(abs(x))) - (n - 1))
    return float(f"{math.ceil(x / f - 1e-9) * f:.12g}")


write = "--write" in sys.argv
violations = []
rows = [json.loads(l) for l in open([a for a in sys.argv[1:] if not a.startswith("--")][0]) if l.strip()]
values = {}
for key, metrics in parity.SPEC.items():
    mine = [r for r in rows if r["test"] == key[0] and r.get("case") == key[1]]
    if not mine:
        print(key, "NO RECORD")
        continue
    values[key] = {}
    for m, spec in metrics.items():
        vals = [r[m] for r in mine]
        slack = spec[1] if len(spec) > 1 else 0.0
        if spec[0] == "max":
            worst = max(vals)
            values[key][m] = sig(FACTOR * worst)
        else:
            worst = min(vals)
            values[key][m] = float(f"{1.0 - sig(FACTOR * (1.0 - worst) + slack):.12g}")
        # the fitted bound is a drift alarm INSIDE the tolerance: never looser than the hand-written, spec-derived ceiling, which this tool
        # never touches - a record that violates a ceiling is a failure of the product, not something to fit a bound around
        ceil = parity.CEILINGS.get(key, {}).get(m)
        if ceil is not None:
            if not parity.holds(spec[0], ceil, worst):
                violations.append((key, m, worst, ceil))
            values[key][m] = min(values[key][m], ceil) if spec[0] == "max" else max(values[key][m], ceil)
        print(f"{str(key):50s} {m:26s} {spec[0]} worst {worst:<12.6g} bound {values[key][m]:<12.6g} ({len(mine)} runs)")
for v in violations:
    print("RECORD VIOLATES THE SPEC-DERIVED CEILING:", v)
if write and violations:
    sys.exit("not writing: fix the product (or argue the ceiling in tests/parity.py by hand), do not fit bounds around a violation")
if write:
    path = os.path.join(ROOT, "tests", "parity.py")
    src = open(path).read()
    body = "VALUES = {\n" + "".join(f"    {k!r}: {v!r},\n" for k, v in values.items()) + "}\n"
    src = re.sub(r"# BEGIN GENERATED VALUES\n.*?# END GENERATED VALUES\n", "# BEGIN GENERATED VALUES\n" + body + "# END GENERATED VALUES\n", src, flags=re.S)
    open(path, "w").write(src)
    print("wrote", path)
