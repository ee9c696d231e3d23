#!/bin/bash
# Round 6: yolo_head_score_kernel after batching its logit loads - plugin / engine tests, then its row in a one-context rocprof pass
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp TRTX_TACTIC_CACHE=/tmp/trtx_tactics.txt
O=$R/gpurun_out/${1:-r06_hs}; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_yolo_plugins.py tests/test_gpu_engine.py tests/test_ref_pinning.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 300 python bench.py --steps 5 --warmup 2 --contexts 1 --no-cpu-baseline --no-tolerance-engine > /dev/null 2>&1
(cd /tmp && TRTX_LANES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --contexts 1 --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-engine > $O/prof.log 2>&1)
python tools/rocprof_summary.py $O/prof > $O/kernel_stats.txt 2>&1; rm -rf $O/prof
grep -E "yolo|total kernel" $O/kernel_stats.txt | cut -c1-170
