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
        if "host_fed" in d and "pcie" in d["host_fed"]:
            h = d["host_fed"]
            print("    host_fed legs", h["legs_ms"], "h2d alone ms", round(h["pcie"]["h2d_alone_ms_per_step"], 4), "GB/s", round(h["pcie"]["h2d_alone_GBps"] or 0, 1), "frames resident ms", round(h["pcie"].get("frames_resident_ms_per_step", 0), 4))
        if "tolerance_engine" in d:
            t = d["tolerance_engine"]
            print("    tolerance_engine", round(t["value"]), "img/s ms", round(t["ms_per_step"], 4), "single", round(t["single_context"]["ms_per_step"], 4), "frac", round(t["roofline"]["frac"] or 0, 3),
                  "parity", {k: v for k, v in (t.get("parity") or {}).items() if k in ("max_logit_err_vs_fp64", "min_iou", "matched", "candidates", "tolerance_met")})
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
