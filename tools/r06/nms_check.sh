#!/bin/bash
# Round 6: the fused NMS kernel - the reference pins and plugin tests, then its duration inside a bench step (rocprofv3 kernel stats, one context)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_nms}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_yolo_plugins.py tests/test_ref_pinning.py tests/test_gpu_yolo5.py tests/test_gpu_yolo8_tasks.py tests/test_gpu_yolo8_branches.py -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest_nms.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python $R/bench.py --steps 30 --warmup 5 --contexts 1 --no-cpu-baseline --no-tolerance-engine > $O/bench_1ctx.json 2> $O/bench_1ctx.err
cd $R; python - <<PY
import csv, glob
f = glob.glob("$O/prof/**/s_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    n = r["Name"]
    if "yolo" in n or "nms" in n:
        print("%-90s calls %5s  avg %8.1f us" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
