export TMPDIR=/tmp
mkdir -p gpurun_out/final2
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/final2/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final2/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_profiles.sh
for m in "resnet50 batch=32 fp16=1" "retinaface_r50 batch=1 fp16=1 h=1280 w=1280" "rcnn_r50c4 batch=1 fp16=1" "rcnn_r50c4 batch=4 fp16=1"; do
  timeout 300 python tools/model_profile.py $m 2>&1 | grep -v amdgpu.ids | head -3
done > gpurun_out/final2/models.log 2>&1; cat gpurun_out/final2/models.log
