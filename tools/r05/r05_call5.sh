#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_f32c; mkdir -p $O; cd $R
timeout 120 tools/hip/bin/mfma_f32_rate 2>&1 | tee $O/mfma_f32_rate.txt
timeout 300 python -m pytest tests/test_gpu_conv_f32.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_conv_f32.txt
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "fp32 or folded_upsample_small" 2>&1 | tail -15 | tee $O/pytest_engine_fp32.txt
timeout 300 python tools/f32_engine_probe.py 2>&1 | grep -v "   tactic" | tee $O/probe.txt
