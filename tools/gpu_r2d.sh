#!/bin/bash
# round-2 call D: compact-epilogue kernels. Full GPU suite, then bench A/B against the previous build (tools/ab/libtrtx_hip_before.so)
set -u
O=gpurun_out/r2d
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
run() { # tag, env..., contexts
  local tag=$1; shift
  local c=$1; shift
  env "$@" timeout 300 python bench.py --contexts $c --no-cpu-baseline --dump-ops $O/ops_$tag.json > $O/bench_$tag.json 2> $O/bench_$tag.err
}
run new_tune_c1 1 TRTX_TUNE_VERBOSE=1
run new_notune_c1 1 TRTX_TUNE=0
run new_tune_c4 4 TRTX_TUNE=1
run new_notune_c4 4 TRTX_TUNE=0
cp tensorrtx_amd/lib/libtrtx_hip.so $O/../libtrtx_hip_new.so.keep 2>/dev/null
cp tools/ab/libtrtx_hip_before.so tensorrtx_amd/lib/libtrtx_hip.so
run old_tune_c1 1 TRTX_TUNE=1
run old_notune_c1 1 TRTX_TUNE=0
run old_notune_c4 4 TRTX_TUNE=0
rm -f $O/../libtrtx_hip_new.so.keep
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2d/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(r["value"]), round(r["ms_per_step"],3), "single", round(r.get("single_context",{}).get("ms_per_step",0),3), "d2h", round(r["d2h_inclusive"]["ms_per_step"],3), "host", round(r["host_fed"]["ms_per_step"],3), "frac", round(r["roofline"]["frac"],4), "avg_us", round(r["roofline"]["avg_launch_us"],2), "all_ms", round(r["roofline"]["all_kernels_ms_per_step"],3), r["roofline"]["tactics"]["moved_off_default"])
    except Exception as e:
        print(f, "ERR", e)
PY
