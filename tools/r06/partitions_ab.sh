#!/bin/bash
# (Kept for the record: needs the trial build with trtx_stream_create_partition + bench.py --partitions, removed after the run - profiles/r06_cu_partitions.txt.)
# Round 6: execution contexts on PARTITIONS of the chip (streams with a CU mask: 32 / P compute units of every XCD each) against contexts sharing the whole chip
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp TRTX_TACTIC_CACHE=/tmp/trtx_tactics.txt
O=$R/gpurun_out/${1:-r06_part}; mkdir -p $O; cd $R
run() {  # label, bench args
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-engine $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json "$1" <<'P' | tee -a $O/summary.txt
import json, sys
ok = False
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line); ok = True
        print(f"{sys.argv[2]:28s} value {r['value']:7.0f} img/s  ms/step {r['ms_per_step']:.4f}  legs {r.get('legs_ms')}")
if not ok: print(sys.argv[2], "FAILED")
P
}
run c3_shared "--contexts 3"
run c2_p2 "--contexts 2 --partitions 2"
run c4_p4 "--contexts 4 --partitions 4"
run c4_p2 "--contexts 4 --partitions 2"
run c8_p8 "--contexts 8 --partitions 8"
run c8_p4 "--contexts 8 --partitions 4"
run c6_p2 "--contexts 6 --partitions 2"
run c3_shared_b "--contexts 3"
tail -3 $O/bench_c4_p4.err
