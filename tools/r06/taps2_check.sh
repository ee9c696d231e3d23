#!/bin/bash
# Round 6: the A-direct kernel's thin 3x3 form (16 input channels, two taps per k-step; ws == 8) - bit-identity of every tactic, per-shape A/B on YOLOv8n's 160 x 160 layers
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_t2}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "every_conv_tactic" 2>&1 | tail -8 | tee $O/pytest_tactics.txt
{ timeout 300 python tools/conv_shape_ab.py 32 160 160 16 16; timeout 300 python tools/conv_shape_ab.py --stride 2 32 320 320 16 32; } 2>&1 | tee $O/taps2_shape_ab.txt
