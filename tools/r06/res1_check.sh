#!/bin/bash
# Round 6: the resident-operand 1x1 kernel (ws == 8) - bit-identity of every tactic, per-shape A/B on the 1x1 layers of YOLOv8n b32.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_res1}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "every_conv_tactic" 2>&1 | tail -8 | tee $O/pytest_tactics.txt
timeout 600 python tools/conv_shape_ab.py --k 1 32 160 160 32 32  32 80 80 64 64  32 80 80 128 64  32 40 40 128 128  32 40 40 256 128  32 20 20 256 256  32 20 20 384 256  32 20 20 512 256  32 40 40 192 128  32 80 80 96 64  32 80 80 80 80 2>&1 | grep -E "GFLOP|res1| ws |igemm  bn +(64|128|32|80) bk 32 bm 128" | tee $O/res1_shape_ab.txt
