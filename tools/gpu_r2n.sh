#!/bin/bash
# round-2 last sanity run of the final build: the default bench command (short) and smoke()
set -u
O=gpurun_out/r2n
mkdir -p $O
timeout 100 python bench.py --no-cpu-baseline --steps 50 > $O/bench_c3.json 2> $O/bench_c3.err
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python - <<'PY'
import json
f="gpurun_out/r2n/bench_c3.json"
try:
    r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r["roofline"]
    print(round(r["value"]), round(r["ms_per_step"],3), "frac", round(rf["frac"],4), "avg_us", round(rf["avg_launch_us"],2), "events", round(rf["avg_launch_us_between_stream_events"],2), "single", round(r["single_context"]["ms_per_step"],3), r["single_context"]["roofline"]["hbm_frac"], "moved", rf["tactics"]["moved_off_default"], rf["tactics"]["default_sum_us"], rf["tactics"]["chosen_sum_us"])
except Exception as e:
    print("ERR",e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
