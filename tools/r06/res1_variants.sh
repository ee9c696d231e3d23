#!/bin/bash
# Round 6: the 1x1 resident kernel's launch geometry (waves per workgroup, row fragments per wave) under three contexts in flight: bench.py per variant
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_res1_var}; mkdir -p $O; cd $R
run() {  # name, TRTX_CONV_RES, TRTX_CONV_DBG
  TRTX_CONV_RES=$2 TRTX_CONV_DBG=$3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-engine > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("%-28s value %6d img/s  ms/step %.4f  single %.4f  frac %.4f" % ("$1", round(d["value"]), d["ms_per_step"], d.get("single_context",{}).get("ms_per_step"), d["roofline"].get("frac")))
PY
}
for rep in 1 2; do
run none_$rep 0 0
run r1_nw16_persist_$rep 2 0
run r1_nw4_f4_$rep 2 $((16 + 4*256))
run r1_nw4_f8_$rep 2 $((16 + 8*256))
run r1_nw4_f16_$rep 2 $((16 + 16*256))
run r1_nw8_f8_$rep 2 $((32 + 8*256))
run r1_nw16_f4_$rep 2 $((0 + 4*256))
done
