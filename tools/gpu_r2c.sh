#!/bin/bash
# round-2 call C: every conv tactic vs torch, multi-context equality, tuned vs untuned bench at 1 / 4 contexts, context sweep
set -u
O=gpurun_out/r2c
mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_multi_context.py -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for c in 4 1; do
  TRTX_TUNE_VERBOSE=1 timeout 300 python bench.py --contexts $c --no-cpu-baseline --dump-ops $O/ops_tune_c$c.json > $O/bench_tune_c$c.json 2> $O/bench_tune_c$c.err
  TRTX_TUNE=0 timeout 300 python bench.py --contexts $c --no-cpu-baseline --dump-ops $O/ops_notune_c$c.json > $O/bench_notune_c$c.json 2> $O/bench_notune_c$c.err
done
for c in 2 3 6; do
  timeout 300 python bench.py --contexts $c --no-cpu-baseline > $O/bench_tune_c$c.json 2> $O/bench_tune_c$c.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2c/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(r["value"]), round(r["ms_per_step"],3), "single", round(r.get("single_context",{}).get("ms_per_step",0),3), "d2h", round(r["d2h_inclusive"]["ms_per_step"],3), "host", round(r["host_fed"]["ms_per_step"],3), "frac", round(r["roofline"]["frac"],4), "avg_us", round(r["roofline"]["avg_launch_us"],2), r["roofline"]["tactics"]["moved_off_default"])
    except Exception as e:
        print(f, "ERR", e)
PY
