#!/bin/bash
export TRTX_OP_REPS=20
for shape in "64 64 3 80 32" "64 80 3 80 32" "32 32 3 80 32" "32 32 1 160 32" "48 32 1 160 32" "64 64 1 80 32" "128 64 1 80 32" "192 64 1 80 32" "96 64 1 80 32" "80 80 1 80 32"; do
  for dbg in 0 4 15; do TRTX_WS_DBG=$dbg python tools/ws_probe.py $shape; done
  TRTX_CONV_NOWS=1 python tools/ws_probe.py $shape
done 2>&1 | grep -v amdgpu.ids
