#!/bin/bash
# Round 6: how much faster (timed ALONE) a candidate has to be before a layer leaves its default kernel - TRTX_TUNE_MARGIN - against `value` (three contexts in flight)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_tm}; mkdir -p $O; cd $R
run() {
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-engine > $O/bench_$1.json 2>/dev/null
  python - $O/bench_$1.json "$1" <<'P' | tee -a $O/summary.txt
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line); rf = r["roofline"]
        print(f"{sys.argv[2]:14s} value {r['value']:7.0f} img/s  ms/step {r['ms_per_step']:.4f}  single {r['single_context']['ms_per_step']:.4f} ms  frac {rf['frac']:.4f}  legs {r.get('legs_ms')}")
P
}
for rep in 1 2; do
  unset TRTX_TUNE TRTX_TUNE_MARGIN
  for m in ${MARGINS:-3 20 30 45 90}; do TRTX_TUNE_MARGIN=$m run margin${m}_$rep; done
  TRTX_TUNE=0 run untuned_$rep
done
