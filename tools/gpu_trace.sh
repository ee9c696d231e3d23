export TMPDIR=/tmp TRTX_CONV_BK32=1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace6 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace6.log 2>&1
cd $GRAFT_REPO_ROOT; F=$(find gpurun_out/trace6 -name "*kernel_trace.csv" | head -1); python tools/trace_overlap.py $F 77 | head -90
