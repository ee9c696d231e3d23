#!/bin/bash
# Round 6, first GPU call: the resident-operand 3x3 kernel (conv_res.hip, tactic ws == 7) - bit-identity against the other tile shapes, then per-shape A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r06_c1; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "every_conv_tactic" 2>&1 | tail -15 | tee $O/pytest_tactics.txt
timeout 400 python tools/conv_shape_ab.py 32 80 80 32 32  32 40 40 64 64  32 20 20 64 64  32 80 80 64 64  32 80 80 64 80 2>&1 | tee $O/res3_shape_ab.txt
