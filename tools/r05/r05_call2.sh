#!/bin/bash
# Round 5, second GPU call: the fp32 MFMA convolution (unit tests, the fp32 engines' parity tests) and the fp32 YOLOv8n engine at C3 against the scalar path.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_f32; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_conv_f32.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_conv_f32.txt
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "fp32 or lenet or explicit" 2>&1 | tail -15 | tee $O/pytest_engine_fp32.txt
timeout 300 python tools/f32_engine_probe.py 2>&1 | tee $O/probe_mfma.txt
timeout 300 python tools/f32_engine_probe.py --direct --steps 3 --contexts 1 --no-oracle 2>&1 | tee $O/probe_direct.txt
timeout 200 python -m pytest tests/test_gpu_multi_context.py -m gpu -q -k rcnn 2>&1 | tail -30 | tee $O/pytest_ctx_rcnn_product.txt
