#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_f32f; mkdir -p $O; cd $R
# tile-count quantisation: 64->64 3x3 with 256 / 512 / 1024 / 2048 / 4096 tiles of 128 x 64 (1, 2, 4, 8, 16 per CU)
timeout 300 python tools/conv_f32_shape_ab.py --only 64,128,1 8 64 64 64 64 3 1 16 64 64 64 64 3 1 32 64 64 64 64 3 1 64 64 64 64 64 3 1 128 64 64 64 64 3 1 2>&1 | grep -v amdgpu | tee $O/tiles_per_cu.txt
timeout 300 python tools/conv_f32_shape_ab.py --only 64,64,1 8 64 64 64 64 3 1 16 64 64 64 64 3 1 32 64 64 64 64 3 1 64 64 64 64 64 3 1 2>&1 | grep -v amdgpu | tee -a $O/tiles_per_cu.txt
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "256x256" 2>&1 | tail -3 | tee $O/pytest_gemm256.txt
