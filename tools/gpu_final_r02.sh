#!/bin/bash
# final check of round 2: the whole GPU suite + smoke() on the final build
set -u
O=gpurun_out/final_r02
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
cp gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
