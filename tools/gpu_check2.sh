export TMPDIR=/tmp
mkdir -p gpurun_out/r1e
timeout 900 python -m pytest tests/test_gpu_yolo_plugins.py tests/test_gpu_engine.py -m gpu -q -x > gpurun_out/r1e/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r1e/pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r1e/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r1e/bench.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r1e/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r1e/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py gpurun_out/r1e/prof | head -24
