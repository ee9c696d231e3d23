#!/bin/bash
# round-2 call J: tuner phase 0 (whole-network palettes): one context, tuned vs untuned, twice
set -u
O=gpurun_out/r2j
mkdir -p $O
run() { local tag=$1; shift; local c=$1; shift
  env "$@" timeout 300 python bench.py --contexts $c --no-cpu-baseline --steps 100 --dump-ops $O/ops_$tag.json > $O/bench_$tag.json 2> $O/bench_$tag.err; }
run tune_c1a 1 TRTX_TUNE=1 TRTX_TUNE_VERBOSE=1
run notune_c1a 1 TRTX_TUNE=0
run tune_c1b 1 TRTX_TUNE=1 TRTX_TUNE_VERBOSE=1
run notune_c1b 1 TRTX_TUNE=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2j/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(r["value"]), round(r["ms_per_step"],3), "d2h", round(r["d2h_inclusive"]["ms_per_step"],3), "frac", round(r["roofline"]["frac"],4), "avg_us", round(r["roofline"]["avg_launch_us"],2), "all", round(r["roofline"]["all_kernels_ms_per_step"],3), r["roofline"]["tactics"]["moved_off_default"], r["roofline"]["tactics"]["default_sum_us"], r["roofline"]["tactics"]["chosen_sum_us"])
    except Exception as e:
        print(f, "ERR", e)
PY
grep palette gpurun_out/r2j/bench_tune_c1a.err gpurun_out/r2j/bench_tune_c1b.err
