#!/bin/bash
# Round 5, third GPU call: fp32 engine with the fp32 stem kernel, fast epilogue, fused fp32 detect tail, folded upsamples; per-shape A/B of the fp32 conv.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_f32b; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_conv_f32.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_conv_f32.txt
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "fp32 or lenet or explicit or resnet50_small or mish" 2>&1 | tail -15 | tee $O/pytest_engine_fp32.txt
timeout 300 python tools/f32_engine_probe.py 2>&1 | tee $O/probe.txt
TRTX_TUNE=0 timeout 300 python tools/f32_engine_probe.py --no-oracle 2>&1 | grep -v "^   op\|tactic" | tee $O/probe_untuned.txt
timeout 600 python tools/conv_f32_shape_ab.py 2>&1 | tee $O/shape_ab.txt
timeout 300 python tools/conv_f32_shape_ab.py --act none 32 80 80 64 64 3 1 32 80 80 64 64 1 1 32 20 20 512 256 1 1 2>&1 | tee $O/shape_ab_noact.txt
