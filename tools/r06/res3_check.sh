#!/bin/bash
# Round 6: the resident-operand 3x3 kernel - bit-identity of every tactic, phase anatomy, per-shape A/B.   usage: res3_check.sh <out-subdir>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_res3}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "every_conv_tactic" 2>&1 | tail -5 | tee $O/pytest_tactics.txt
timeout 120 tools/hip/bin/res3_anatomy 2>&1 | tee $O/res3_anatomy.txt | grep -A10 "event interval" | grep -v "probe [123]" | grep -B1 -A9 "^64->64 3x3 @80 b32: \|ablate 4\|@40\|32->32" | cut -c1-200 | head -70
timeout 400 python tools/conv_shape_ab.py 32 80 80 32 32  32 40 40 64 64  32 20 20 64 64  32 80 80 64 64  32 80 80 64 80 2>&1 | grep -E "GFLOP|res3|patch|ws " | tee $O/res3_shape_ab.txt
