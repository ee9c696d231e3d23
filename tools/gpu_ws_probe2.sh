export TRTX_OP_REPS=20
for shape in "64 64 3 320 8" "64 64 1 320 8" "64 64 3 200 4" "64 64 1 200 4"; do
  python tools/ws_probe.py $shape; TRTX_CONV_NOWS=1 python tools/ws_probe.py $shape
done 2>&1 | grep -v amdgpu
