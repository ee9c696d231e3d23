#!/bin/bash
# round-2 call M (last): per-launch kernel timestamps in the profile pass (hipExtLaunchKernelGGL start / stop events)
set -u
O=gpurun_out/r2m
mkdir -p $O
timeout 120 python bench.py --contexts 1 --no-cpu-baseline --steps 30 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 120 python bench.py --no-cpu-baseline --steps 50 > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json
for f in ("gpurun_out/r2m/bench_c1.json","gpurun_out/r2m/bench_c3.json"):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rf=r["roofline"]
        print(f.split("/")[-1], round(r["value"]), round(r["ms_per_step"],3), "frac", round(rf["frac"],4), "avg_us", round(rf["avg_launch_us"],2), "events", round(rf["avg_launch_us_between_stream_events"],2), "rocprof", rf["rocprofv3_avg_launch_us"], rf["timing"][:40], "single", r.get("single_context",{}).get("roofline",{}).get("hbm_frac"))
    except Exception as e:
        print(f,"ERR",e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
