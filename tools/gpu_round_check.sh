export TMPDIR=/tmp
mkdir -p gpurun_out/r1c
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r1c/pytest.log 2>&1; echo "pytest rc=$?" 
timeout 300 python bench.py > gpurun_out/r1c/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r1c/bench.log
timeout 120 python tools/nms_probe.py > gpurun_out/r1c/nms.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r1c/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r1c/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/r1c/pytest.log
