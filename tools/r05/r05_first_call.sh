#!/bin/bash
# Round 5, first GPU call: the two operand-reuse kernels written in rounds 2-4 run at last, trimmed to fit ~15 GPU minutes:
# resident-patch (tools/scratch/patch) and row-reuse (tools/scratch/r3) builds against the product library on ONE box.  Results under gpurun_out/r05_first/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/r05_first; mkdir -p $O; cd $R
LP=$R/tools/scratch/patch LR=$R/tools/scratch/r3
# --- resident-patch kernel: bit-identity among the tactics, per-shape A/B
export TRTX_HIP_LIB=$LP/libtrtx_hip.so TRTX_CONV_PATCH=1
timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -5 | tee $O/patch_pytest_conv.txt
timeout 240 python tools/conv_shape_ab.py 2>&1 | tee $O/patch_shape_ab.txt
TRTX_CONV_PATCH=2 timeout 200 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "group" 2>&1 | tail -3 | tee $O/patch_pytest_groups.txt
timeout 200 python -m pytest tests/test_gpu_multi_context.py -m gpu -q 2>&1 | tail -3 | tee $O/patch_pytest_ctx.txt
# --- row-reuse kernel: forced onto every layer it can take under three contexts, then among the tactics
export TRTX_HIP_LIB=$LR/libtrtx_hip.so; unset TRTX_CONV_PATCH
for st in 1 2; do
  TRTX_TUNE=0 TRTX_GROUP_CONVS=0 TRTX_FORCE_R3=$st timeout 120 python tools/coscheduling_bisect.py 8 2>&1 | grep "serial-repeatable" | sed "s/^/forced stages=$((4 - st)): /"
done | tee $O/r3_forced.txt
TRTX_TUNE=0 TRTX_GROUP_CONVS=0 TRTX_FORCE_R3=1 timeout 120 python tools/coscheduling_bisect.py 20 poison 2>&1 | grep "serial-repeatable" | sed "s/^/forced, 20 rounds, poisoned LDS: /" | tee -a $O/r3_forced.txt
TRTX_TACTICS_R3=1 timeout 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_multi_context.py -m gpu -q 2>&1 | tail -3 | tee $O/r3_pytest.txt
# --- bench A/B on this box, alternating
for round in 1 2; do
  for v in product patch patch_groups r3; do
    unset TRTX_HIP_LIB TRTX_CONV_PATCH TRTX_TACTICS_R3
    case $v in
      patch) export TRTX_HIP_LIB=$LP/libtrtx_hip.so TRTX_CONV_PATCH=1;;
      patch_groups) export TRTX_HIP_LIB=$LP/libtrtx_hip.so TRTX_CONV_PATCH=2;;
      r3) export TRTX_HIP_LIB=$LR/libtrtx_hip.so TRTX_TACTICS_R3=1;;
    esac
    TRTX_TACTIC_CACHE=/tmp/tc_$v.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err
  done
done
unset TRTX_HIP_LIB TRTX_CONV_PATCH TRTX_TACTICS_R3
python tools/show_bench.py $O/bench_*.json | tee $O/bench.txt
