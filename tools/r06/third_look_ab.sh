#!/bin/bash
# Round 6: the tuner's third look on (default) / off (TRTX_TUNE=2), same box, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp TRTX_TUNE_VERBOSE=1
O=$R/gpurun_out/${1:-r06_tl2}; mkdir -p $O; cd $R
run() {  # label, args
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-engine $2 > $O/bench_$1.json 2> $O/bench_$1.err
  grep "three contexts in flight" $O/bench_$1.err | sed "s/^/   $1: /" | cut -c1-230 | tee -a $O/summary.txt
  python - $O/bench_$1.json "$1" <<'P' | tee -a $O/summary.txt
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line); rf = r["roofline"]
        print(f"{sys.argv[2]:14s} value {r['value']:8.1f} img/s  ms/step {r['ms_per_step']:.4f}  single {r['single_context']['ms_per_step']:.4f}  frac {rf['frac']:.4f}")
P
}
for rep in 1 2 3; do
  unset TRTX_TUNE; run c3_look_$rep "${ARGS:-}"
  TRTX_TUNE=2 run c3_alone_$rep "${ARGS:-}"
done
