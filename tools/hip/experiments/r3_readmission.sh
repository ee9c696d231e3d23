#!/bin/bash
# The row-reuse 3x3 kernel ("r3", kernels/conv_igemm.hip) back as a tactic - the checks its FIXED build has to pass first.
#   bash tools/r3_readmission.sh build      here (CPU): a second libtrtx_hip.so with -DTRTX_EXPERIMENTAL_R3 under tools/scratch/r3/ (git-ignored, travels with gpurun)
#   gpurun --timeout 900 -- bash tools/r3_readmission.sh run      on the MI355X box: results under gpurun_out/r3_readmission/
# run: (1) the ISA scan (no LDS read in flight at any barrier), (2) the kernel forced onto every layer it can take, three and two stages, three contexts in flight,
# 8 rounds, three times each, plain and with poisoned LDS: every run must say IDENTICAL, (3) tests/test_gpu_conv.py + tests/test_gpu_multi_context.py with the kernel
# among the tuner's candidates (TRTX_TACTICS_R3=1), (4) bench.py A/B on this box: product library / this library with TRTX_TACTICS_R3=1, two alternating runs each,
# grouped launches on and off (the kernel is not a group member yet: if it wins only ungrouped, a grouped entry point is the next step).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/tools/scratch/r3
if [ "${1:-}" = build ]; then
  mkdir -p $L
  cd $R/tensorrtx_amd/csrc
  make -j8 > /dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form \
      -DTRTX_EXPERIMENTAL_R3 -c kernels/conv_igemm.hip -o $L/conv_igemm_r3.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -mllvm -amdgpu-mfma-vgpr-form -DTRTX_EXPERIMENTAL_R3 -S --cuda-device-only \
      -o $L/conv_igemm_r3.s kernels/conv_igemm.hip 2> /dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(find build -name "*.o" ! -name "conv_igemm.o") $L/conv_igemm_r3.o -o $L/libtrtx_hip.so || exit 1
  python $R/tools/isa_barrier_reads.py $L/conv_igemm_r3.s
  ls -la $L
  exit 0
fi
export TMPDIR=/tmp
O=$R/gpurun_out/r3_readmission; mkdir -p $O; cd $R
python tools/isa_barrier_reads.py $L/conv_igemm_r3.s > $O/isa_scan.txt 2>&1; tail -1 $O/isa_scan.txt
export TRTX_HIP_LIB=$L/libtrtx_hip.so
for st in 1 2; do for rep in 1 2 3; do
  TRTX_TUNE=0 TRTX_GROUP_CONVS=0 TRTX_FORCE_R3=$st timeout 120 python tools/coscheduling_bisect.py 8 2>&1 | grep "serial-repeatable" | sed "s/^/forced stages=$((4 - st)) run $rep: /"
done; done | tee $O/forced.txt
TRTX_TUNE=0 TRTX_GROUP_CONVS=0 TRTX_FORCE_R3=1 timeout 120 python tools/coscheduling_bisect.py 20 poison 2>&1 | grep "serial-repeatable" | sed "s/^/forced, 20 rounds, poisoned LDS: /" | tee -a $O/forced.txt
TRTX_TACTICS_R3=1 timeout 400 python -m pytest tests/test_gpu_conv.py tests/test_gpu_multi_context.py tests/test_gpu_tactics.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
# VERDICT r3 item 3's bar: 20 consecutive green runs of the three-context test with the kernel among the tactics
for i in $(seq 1 20); do TRTX_TACTICS_R3=1 timeout 60 python -m pytest tests/test_gpu_multi_context.py -m gpu -q -k "yolov8n_three_contexts_in_flight" 2>&1 | tail -1; done | sort | uniq -c | tee $O/twenty_runs.txt
for round in 1 2; do
  for v in product r3 r3_ungrouped; do
    case $v in
      product) unset TRTX_HIP_LIB TRTX_TACTICS_R3 TRTX_GROUP_CONVS;;
      r3) export TRTX_HIP_LIB=$L/libtrtx_hip.so TRTX_TACTICS_R3=1; unset TRTX_GROUP_CONVS;;
      r3_ungrouped) export TRTX_HIP_LIB=$L/libtrtx_hip.so TRTX_TACTICS_R3=1 TRTX_GROUP_CONVS=0;;
    esac
    TRTX_TACTIC_CACHE=/tmp/tc_$v.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${v}_$round.json 2> /dev/null
  done
done
unset TRTX_HIP_LIB TRTX_TACTICS_R3 TRTX_GROUP_CONVS
python tools/show_bench.py $O/bench_*.json | tee $O/bench.txt
