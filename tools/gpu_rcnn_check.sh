export TMPDIR=/tmp
mkdir -p gpurun_out/r1d
timeout 1200 python -m pytest tests/test_gpu_rcnn.py -m gpu -q -x > gpurun_out/r1d/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r1d/pytest.log
timeout 300 python tools/model_profile.py rcnn_r50c4 batch=1 fp16=1 > gpurun_out/r1d/rcnn_prof.log 2>&1; tail -30 gpurun_out/r1d/rcnn_prof.log
