#!/bin/bash
# final check of round 2, second half: the GPU test files after the point where the first run stopped (-x on a test-only assertion)
set -u
O=gpurun_out/final_r02
mkdir -p $O
timeout 175 python -m pytest tests/test_gpu_tactics.py tests/test_gpu_yolo5.py tests/test_gpu_yolo8_branches.py tests/test_gpu_yolo8_tasks.py tests/test_gpu_yolo_plugins.py tests/test_oracle_det.py tests/test_oracle_yolo.py tests/test_ref_pinning.py tests/test_replicas_gloo.py tests/test_runtime_cpu.py -m gpu -q > $O/pytest_part2.log 2>&1
tail -6 $O/pytest_part2.log
