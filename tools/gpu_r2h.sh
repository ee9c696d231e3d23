#!/bin/bash
# round-2 call H: MFMA accumulators kept in VGPRs (-mllvm -amdgpu-mfma-vgpr-form): conv tests, then A/B against the previous build
set -u
O=gpurun_out/r2h
mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_int8.py -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
run() { local tag=$1; shift; local c=$1; shift
  env "$@" timeout 300 python bench.py --contexts $c --no-cpu-baseline --steps 100 --dump-ops $O/ops_$tag.json > $O/bench_$tag.json 2> $O/bench_$tag.err; }
run new_c1 1 TRTX_TUNE=1
run new_c3 3 TRTX_TUNE=1
run new_notune_c1 1 TRTX_TUNE=0
run new_notune_c3 3 TRTX_TUNE=0
cp tools/ab/libtrtx_hip_before.so tensorrtx_amd/lib/libtrtx_hip.so
run old_c1 1 TRTX_TUNE=1
run old_c3 3 TRTX_TUNE=1
run old_notune_c1 1 TRTX_TUNE=0
run old_notune_c3 3 TRTX_TUNE=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2h/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        sc=r.get("single_context",{})
        print(f.split("/")[-1], round(r["value"]), round(r["ms_per_step"],3), "single", round(sc.get("ms_per_step",0),3), "single_frac", round(sc.get("roofline",{}).get("hbm_frac",0),4), "d2h", round(r["d2h_inclusive"]["ms_per_step"],3), "host", round(r["host_fed"]["ms_per_step"],3), "frac", round(r["roofline"]["frac"],4), "avg_us", round(r["roofline"]["avg_launch_us"],2), "all", round(r["roofline"]["all_kernels_ms_per_step"],3), r["roofline"]["tactics"]["moved_off_default"])
    except Exception as e:
        print(f, "ERR", e)
PY
