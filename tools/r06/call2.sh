#!/bin/bash
# Round 6: the resident-operand 3x3 kernel after the first anatomy - pinned scalars, pipelined k-loop: bit-identity, anatomy, per-shape A/B
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r06_c3; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "every_conv_tactic" 2>&1 | tail -5 | tee $O/pytest_tactics.txt
timeout 120 tools/hip/bin/res3_anatomy 2>&1 | tee $O/res3_anatomy.txt | grep -A12 "event interval" | grep -v "probe [123]" | head -80
timeout 400 python tools/conv_shape_ab.py 32 80 80 32 32  32 40 40 64 64  32 20 20 64 64  32 80 80 64 64  32 80 80 64 80 2>&1 | grep -E "GFLOP|res3|patch|ws " | tee $O/res3_shape_ab.txt
