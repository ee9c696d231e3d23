#!/bin/bash
# Round 6: the engine with / without the resident-operand 3x3 kernel (TRTX_CONV_RES=0): engine-level bit tests, then bench.py alternating on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_engine_ab}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_multi_context.py tests/test_gpu_tactics.py -m gpu -q -x -k "not fp32 and not rcnn and not retina and not resnet" 2>&1 | tail -8 | tee $O/pytest_engine.txt
for rep in 1 2; do
  for res in 7 0; do
    TRTX_CONV_RES=$res TRTX_TUNE_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-engine > $O/bench_res${res}_$rep.json 2> $O/bench_res${res}_$rep.err
    python - <<PY
import json
d=json.loads(open("$O/bench_res${res}_$rep.json").read().strip().splitlines()[-1])
print("res=$res rep=$rep value", round(d["value"]), "img/s  ms/step", d["ms_per_step"], " single", d.get("single_context",{}).get("ms_per_step"), " frac", d["roofline"].get("frac"))
PY
    grep -c "res3" $O/bench_res${res}_$rep.err
  done
done
