#!/bin/bash
# round-2 call K: palettes for throughput engines (3 contexts): tuned vs untuned, twice
set -u
O=gpurun_out/r2k
mkdir -p $O
run() { local tag=$1; shift; local c=$1; shift
  env "$@" timeout 300 python bench.py --contexts $c --no-cpu-baseline --steps 100 --dump-ops $O/ops_$tag.json > $O/bench_$tag.json 2> $O/bench_$tag.err; }
run tune_c3a 3 TRTX_TUNE=1 TRTX_TUNE_VERBOSE=1
run notune_c3a 3 TRTX_TUNE=0
run tune_c3b 3 TRTX_TUNE=1 TRTX_TUNE_VERBOSE=1
run notune_c3b 3 TRTX_TUNE=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2k/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        sc=r.get("single_context",{})
        print(f.split("/")[-1], round(r["value"]), round(r["ms_per_step"],3), "single", round(sc.get("ms_per_step",0),3), "single_frac", round(sc.get("roofline",{}).get("hbm_frac",0),4), "d2h", round(r["d2h_inclusive"]["ms_per_step"],3), "host", round(r["host_fed"]["ms_per_step"],3), "frac", round(r["roofline"]["frac"],4), "avg_us", round(r["roofline"]["avg_launch_us"],2), r["roofline"]["tactics"]["moved_off_default"], r["roofline"]["tactics"]["default_sum_us"], r["roofline"]["tactics"]["chosen_sum_us"])
    except Exception as e:
        print(f, "ERR", e)
PY
grep -h palette gpurun_out/r2k/bench_tune_c3a.err | head -12
