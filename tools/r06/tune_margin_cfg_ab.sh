#!/bin/bash
# Round 6: TRTX_TUNE_MARGIN 3 vs 30 on the other configurations, same box, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_tm3}; mkdir -p $O; cd $R
run() {  # label, args
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-engine $2 > $O/bench_$1.json 2>/dev/null
  python - $O/bench_$1.json "$1" <<'P' | tee -a $O/summary.txt
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line); rf = r["roofline"]
        print(f"{sys.argv[2]:22s} value {r['value']:8.1f} img/s  ms/step {r['ms_per_step']:.4f}  frac {rf['frac']:.4f}")
P
}
for rep in 1 2; do
  for m in 3 30; do
    TRTX_TUNE_MARGIN=$m run c2_m${m}_$rep "--config resnet50"
    TRTX_TUNE_MARGIN=$m run int8_m${m}_$rep "--precision int8"
    TRTX_TUNE_MARGIN=$m run c4_m${m}_$rep "--config retinaface_r50"
  done
done
