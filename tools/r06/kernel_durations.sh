#!/bin/bash
# Round 6: rocprofv3 kernel durations (dispatch begin -> end) of the tactics of a few layers, input flushed between launches (tools/conv_shape_ab.py)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_kdur}; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p3 -o s -- python $R/tools/conv_shape_ab.py 32 80 80 64 64  32 40 40 64 64  32 80 80 32 32 > $O/ab3.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -o s -- python $R/tools/conv_shape_ab.py --k 1 32 80 80 64 64  32 80 80 128 64  32 40 40 256 128  32 20 20 384 256 > $O/ab1.txt 2>&1
cd $R
python - <<PY
import csv, glob
for tag in ("p3", "p1"):
    f = glob.glob("$O/%s/**/s_kernel_stats.csv" % tag, recursive=True)
    print("==", tag)
    for r in csv.DictReader(open(f[0])):
        n = r["Name"]
        if "conv" in n:
            print("%-110s calls %5s  avg %8.1f us  min %8.1f  max %8.1f" % (n[:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
