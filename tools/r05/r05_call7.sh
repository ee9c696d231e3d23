#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_f32e; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv_f32.py tests/test_gpu_engine.py -m gpu -q -k "f32 or fp32 or folded_upsample_small" 2>&1 | tail -8 | tee $O/pytest.txt
timeout 300 python tools/f32_engine_probe.py 2>&1 | grep -v "   tactic" | tee $O/probe.txt
bash tools/pmc_conv_f32.sh "32 80 80 64 64 3 1" "64,128,1" c64 2>&1 | tee $O/pmc_c64.txt
bash tools/pmc_conv_f32.sh "32 40 40 384 128 1 1" "128,128,1" p384 2>&1 | tee $O/pmc_p384.txt
