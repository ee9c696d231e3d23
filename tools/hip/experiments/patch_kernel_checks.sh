#!/bin/bash
# The resident-patch 3x3 kernel (kernels/conv_igemm.hip, -DTRTX_EXPERIMENTAL_PATCH) was written at the end of round 4 WITHOUT GPU minutes:
# its index arithmetic is replayed on the CPU (tests/test_patch_index_cpu.py), its ISA is scanned (tools/isa_barrier_reads.py), it has never run.
#   bash tools/patch_kernel_checks.sh build     here (CPU): a second libtrtx_hip.so with the define under tools/scratch/patch/ (git-ignored, travels with gpurun)
#   gpurun --timeout 900 -- bash tools/patch_kernel_checks.sh run     on the MI355X box: results under gpurun_out/patch_kernel/
# run: (1) tests/test_gpu_conv.py with the kernel among every eligible layer's tactics (TRTX_CONV_PATCH=1: the tactic test asserts it BIT-IDENTICAL to the
# main kernel's tiles), (2) the multi-context tests with it among the tuner's candidates, (3) per-layer timings of the YOLOv8n head shapes, kernel forced vs
# default (tools/conv_shape_ab.py prints both), (4) bench.py A/B on this box: product library / this library with TRTX_CONV_PATCH=1, grouped launches on / off
# (the kernel is not a group member), two alternating runs each.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/tools/scratch/patch
if [ "${1:-}" = build ]; then
  mkdir -p $L
  cd $R/tensorrtx_amd/csrc
  make -j8 > /dev/null || exit 1
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -DTRTX_EXPERIMENTAL_PATCH"
  /opt/rocm/bin/hipcc $FLAGS -c kernels/conv_igemm.hip -o $L/conv_igemm_patch.o || exit 1
  /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o $L/conv_igemm_patch.s kernels/conv_igemm.hip 2> /dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(find build -name "*.o" ! -name "conv_igemm.o") $L/conv_igemm_patch.o -o $L/libtrtx_hip.so || exit 1
  python $R/tools/isa_barrier_reads.py $L/conv_igemm_patch.s
  rm -f $L/conv_igemm_patch.s
  ls -la $L
  exit 0
fi
export TMPDIR=/tmp
O=$R/gpurun_out/patch_kernel; mkdir -p $O; cd $R
export TRTX_HIP_LIB=$L/libtrtx_hip.so TRTX_CONV_PATCH=1
timeout 400 python -m pytest tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_conv.txt
timeout 300 python -m pytest tests/test_gpu_multi_context.py tests/test_gpu_tactics.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest_ctx.txt
timeout 300 python tools/conv_shape_ab.py 2>&1 | tee $O/shape_ab.txt
# grouped launches on the patch kernel against the same engine with the groups on the main kernel: every output bit for bit (one context, then three in flight)
TRTX_CONV_PATCH=2 timeout 200 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "group" 2>&1 | tail -3 | tee $O/pytest_groups.txt
for round in 1 2; do
  for v in product patch patch_groups patch_ungrouped; do
    case $v in
      product) unset TRTX_HIP_LIB TRTX_CONV_PATCH TRTX_GROUP_CONVS;;
      patch) export TRTX_HIP_LIB=$L/libtrtx_hip.so TRTX_CONV_PATCH=1; unset TRTX_GROUP_CONVS;;
      patch_groups) export TRTX_HIP_LIB=$L/libtrtx_hip.so TRTX_CONV_PATCH=2; unset TRTX_GROUP_CONVS;;   # ... and the grouped launches whose members all qualify (the head's second 3x3s)
      patch_ungrouped) export TRTX_HIP_LIB=$L/libtrtx_hip.so TRTX_CONV_PATCH=1 TRTX_GROUP_CONVS=0;;
    esac
    TRTX_TACTIC_CACHE=/tmp/tc_$v.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${v}_$round.json 2> /dev/null
  done
done
unset TRTX_HIP_LIB TRTX_CONV_PATCH TRTX_GROUP_CONVS
python tools/show_bench.py $O/bench_*.json | tee $O/bench.txt
