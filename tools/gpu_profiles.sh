# end-of-round evidence: headline bench line, rocprofv3 kernel stats of the same command, PMC HBM traffic of the conv kernels
export TMPDIR=/tmp
mkdir -p gpurun_out/evidence
timeout 600 python bench.py > gpurun_out/evidence/bench.log 2>&1; tail -1 gpurun_out/evidence/bench.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/evidence/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/evidence/prof.log 2>&1)
python tools/rocprof_summary.py gpurun_out/evidence/prof > gpurun_out/evidence/kernel_stats.txt; head -24 gpurun_out/evidence/kernel_stats.txt
bash tools/pmc_bench_traffic.sh
python tools/layer_table.py gpurun_out/evidence/layers.json > gpurun_out/evidence/layers.txt 2>&1; tail -1 gpurun_out/evidence/layers.txt
