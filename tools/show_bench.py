"""One-line digest of bench.py JSON lines: python tools/show_bench.py file.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "legs", d.get("legs_ms"), "single", round(d.get("single_context", {}).get("ms_per_step", 0), 4),
              "d2h", round(d.get("d2h_inclusive", {}).get("value", 0)), "host_fed", round(d.get("host_fed", {}).get("value", 0)), "conv_ms", round(r["conv_ms_per_step"], 4),
              "launches", r["launches_per_step"], "frac", round(r["frac"], 3), "suspect", d.get("suspect"))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
