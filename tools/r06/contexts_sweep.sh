#!/bin/bash
# Round 6: execution contexts in flight on the bench line with the round's kernels (2..6), same box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp TRTX_TACTIC_CACHE=/tmp/trtx_tactics.txt
O=$R/gpurun_out/${1:-r06_ctx}; mkdir -p $O; cd $R
for c in 3 2 4 5 6 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --contexts $c --no-cpu-baseline --no-tolerance-engine > $O/bench_c$c.json 2> $O/bench_c$c.err
  python - $O/bench_c$c.json $c <<'P' | tee -a $O/summary.txt
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line)
        print(f"contexts {sys.argv[2]}: value {r['value']:.0f} img/s  ms/step {r['ms_per_step']:.4f}  legs {r.get('legs_ms')}")
P
done
