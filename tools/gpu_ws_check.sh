#!/bin/bash
# numerics of the conv kernels + A/B timing weight-stationary vs implicit GEMM on the YOLOv8n shapes + the engine step
mkdir -p gpurun_out/ws
python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -15 > gpurun_out/ws/pytest_conv.log; tail -3 gpurun_out/ws/pytest_conv.log
python tools/conv_ab.py ws > gpurun_out/ws/ab_ws.log 2>&1; tail -2 gpurun_out/ws/ab_ws.log
TRTX_WS_OCC=1 python tools/conv_ab.py ws_occ1 > gpurun_out/ws/ab_ws1.log 2>&1; tail -2 gpurun_out/ws/ab_ws1.log
TRTX_CONV_NOWS=1 python tools/conv_ab.py igemm > gpurun_out/ws/ab_igemm.log 2>&1; tail -2 gpurun_out/ws/ab_igemm.log
python bench.py --no-cpu-baseline --steps 30 > gpurun_out/ws/bench_ws.log 2>&1; tail -1 gpurun_out/ws/bench_ws.log | cut -c1-330
