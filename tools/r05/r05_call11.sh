#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_f32i; mkdir -p $O; cd $R
timeout 600 python tools/conv_f32_shape_ab.py --only 64,128,1,16 32 80 80 64 64 3 1 32 40 40 64 64 3 1 128 64 64 64 64 3 1 128 40 40 512 64 3 1 2>&1 | grep -v amdgpu | tee $O/setprio.txt
timeout 600 python tools/conv_f32_shape_ab.py --only 64,64,1,16 32 80 80 64 64 3 1 32 40 40 64 64 3 1 2>&1 | grep -v amdgpu | tee -a $O/setprio.txt
timeout 300 python tools/f32_engine_probe.py --no-oracle 2>&1 | grep -v "^   op" | tee $O/probe.txt
