#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_f32h; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_conv_f32.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_conv_f32.txt
timeout 600 python tools/conv_f32_shape_ab.py 32 80 80 64 64 3 1 32 40 40 128 64 3 1 32 40 40 64 64 3 1 32 20 20 128 128 3 1 32 80 80 128 64 1 1 32 40 40 384 128 1 1 32 20 20 512 256 1 1 32 80 80 64 64 1 1 2>&1 | grep -v amdgpu | tee $O/shape_ab_bk32.txt
timeout 300 python tools/f32_engine_probe.py 2>&1 | grep -v "^   op" | tee $O/probe.txt
