#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_f32d; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_conv_f32.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_conv_f32.txt
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "fp32 or folded_upsample_small" 2>&1 | tail -15 | tee $O/pytest_engine_fp32.txt
timeout 300 python tools/f32_engine_probe.py 2>&1 | grep -v "   tactic" | tee $O/probe.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python tools/show_bench.py $O/bench.json
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
t=d.get("tolerance_engine")
print(json.dumps(t, indent=1)[:3000] if t else "no tolerance_engine")
print(json.dumps(d.get("parity"), indent=1)[:1500])
PY
