#!/bin/bash
# Where the fp32 MFMA convolution's time goes (it sits at 0.5-0.6 of the fp32 MFMA peak whatever the tile shape): the k-loop with its loads range-checked
# away (TRTX_CONV_DBG=3: MFMAs + fragment reads + barriers only), without its MFMAs (4: the fill path alone), without the epilogue (8), and with 4 / 6 LDS stages.
#   bash tools/f32_ablation.sh build     here: tools/scratch/ablate/libtrtx_hip.so (conv_igemm_f32.hip with -DTRTX_CONV_ABLATE)
#   gpurun -- bash tools/f32_ablation.sh run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/tools/scratch/ablate
if [ "${1:-}" = build ]; then
  mkdir -p $L; cd $R/tensorrtx_amd/csrc; make -j8 > /dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DTRTX_CONV_ABLATE \
      -c kernels/conv_igemm_f32.hip -o $L/conv_igemm_f32_ablate.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(find build -name "*.o" ! -name "conv_igemm_f32.o") $L/conv_igemm_f32_ablate.o -o $L/libtrtx_hip.so || exit 1
  rm -f $L/*.o; ls -la $L; exit 0
fi
export TMPDIR=/tmp TRTX_HIP_LIB=$L/libtrtx_hip.so
O=$R/gpurun_out/r05_f32_ablation; mkdir -p $O; cd $R
SH="32 80 80 64 64 3 1 32 40 40 64 64 3 1 32 20 20 128 64 3 1"
for nst in 0 4 6; do for dbg in 0 3 4 8 7; do
  echo "== stages ${nst/0/3}  TRTX_CONV_DBG=$dbg"
  TRTX_F32_NST=$nst TRTX_CONV_DBG=$dbg timeout 120 python tools/conv_f32_shape_ab.py --only 64,128,1 $SH 2>&1 | grep -v amdgpu.ids | sed 's/same bits//; s/DIFFERS.*//'
done; done | tee $O/ablation.txt
