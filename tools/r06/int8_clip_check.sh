#!/bin/bash
# Round 6: INT8 entropy calibration with the clip limit (TRTX_INT8_CLIP_LIMIT, default 1e-4) against the plain KL threshold (0): detection-level rows of tests/test_gpu_int8.py
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_i8}; mkdir -p $O; cd $R
for lim in 1e-4 0 1e-3 1e-5; do
  rm -f gpurun_out/parity_metrics.jsonl
  TRTX_INT8_CLIP_LIMIT=$lim TRTX_PARITY_DRIFT=warn timeout 1500 python -m pytest tests/test_gpu_int8.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest_$lim.txt
  cp gpurun_out/parity_metrics.jsonl $O/parity_$lim.jsonl
  echo "== limit $lim"; grep -E "int8" $O/parity_$lim.jsonl | grep -v drift | cut -c1-330
done 2>&1 | tee $O/summary.txt
