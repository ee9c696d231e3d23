#!/bin/bash
# round-2 call L: two-stage row-reuse kernel as a tactic: tactic tests, one context tuned twice, three contexts once
set -u
O=gpurun_out/r2l
mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "tactic" > $O/pytest.log 2>&1
tail -2 $O/pytest.log
run() { local tag=$1; shift; local c=$1; shift
  env "$@" timeout 300 python bench.py --contexts $c --no-cpu-baseline --steps 100 --dump-ops $O/ops_$tag.json > $O/bench_$tag.json 2> $O/bench_$tag.err; }
run tune_c1a 1 TRTX_TUNE_VERBOSE=1
run tune_c1b 1 TRTX_TUNE_VERBOSE=1
run tune_c3a 3 TRTX_TUNE_VERBOSE=1
python - <<'PY'
import json,glob,collections
for f in sorted(glob.glob("gpurun_out/r2l/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        sc=r.get("single_context",{})
        print(f.split("/")[-1], round(r["value"]), round(r["ms_per_step"],3), "single", round(sc.get("ms_per_step",0),3), "single_frac", round(sc.get("roofline",{}).get("hbm_frac",0),4), "d2h", round(r["d2h_inclusive"]["ms_per_step"],3), "frac", round(r["roofline"]["frac"],4), "avg_us", round(r["roofline"]["avg_launch_us"],2), r["roofline"]["tactics"]["moved_off_default"], r["roofline"]["tactics"]["default_sum_us"], r["roofline"]["tactics"]["chosen_sum_us"])
    except Exception as e:
        print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r2l/ops_*.tactics.json")):
    c=collections.Counter(t["tactic"] for t in json.load(open(f)))
    print(f.split("/")[-1], dict(c))
PY
