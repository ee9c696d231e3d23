#!/bin/bash
# Round 6: the tuner's third look (three scratch contexts side by side: winners timed alone / large winners only / defaults) - what it decides per configuration, and `value`
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp TRTX_TUNE_VERBOSE=1
O=$R/gpurun_out/${1:-r06_tl}; mkdir -p $O; cd $R
run() {  # label, args
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  grep "three contexts in flight" $O/bench_$1.err | sed "s/^/   $1: /" | tee -a $O/summary.txt
  python - $O/bench_$1.json "$1" <<'P' | tee -a $O/summary.txt
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line); rf = r["roofline"]; t = r.get("tolerance_engine")
        print(f"{sys.argv[2]:10s} value {r['value']:8.1f} img/s  ms/step {r['ms_per_step']:.4f}  single {r['single_context']['ms_per_step']:.4f}  frac {rf['frac']:.4f}" + (f"  | fp32 engine {t['value']:.0f} img/s frac {t['roofline']['frac']:.3f}" if t else ""))
P
}
for rep in 1 2; do
  run c3_$rep ""
  run int8_$rep "--precision int8 --no-tolerance-engine"
  run c2_$rep "--config resnet50"
  run c4_$rep "--config retinaface_r50"
done
run c5 "--config rcnn_r50c4"
