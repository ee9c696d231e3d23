#!/bin/bash
mkdir -p gpurun_out/graph
python -m pytest tests/test_gpu_engine.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline --steps 50 > gpurun_out/graph/bench_graph.log 2>&1; tail -1 gpurun_out/graph/bench_graph.log | cut -c1-330
TRTX_GRAPH=0 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/graph/bench_eager.log 2>&1; tail -1 gpurun_out/graph/bench_eager.log | cut -c1-330
TRTX_LANES=1 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/graph/bench_graph_1lane.log 2>&1; tail -1 gpurun_out/graph/bench_graph_1lane.log | cut -c100-330
TRTX_CONV_NOWS=1 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/graph/bench_graph_nows.log 2>&1; tail -1 gpurun_out/graph/bench_graph_nows.log | cut -c100-330
python tools/split_batch_probe.py 2>&1 | tail -3
