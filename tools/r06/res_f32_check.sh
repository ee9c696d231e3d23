#!/bin/bash
# Round 6: the resident-operand kernels with fp32 operands - every tile shape the same bits, per-shape A/B against the fp32 MFMA peak, the fp32 engine with / without
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_res_f32}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv_f32.py -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest_f32.txt
timeout 600 python tools/conv_f32_shape_ab.py 32 80 80 64 64 3 1  32 40 40 64 64 3 1  32 80 80 32 32 3 1  32 160 160 16 16 3 1  32 80 80 64 80 3 1  32 80 80 80 80 3 1  32 80 80 128 64 1 1  32 80 80 64 64 1 1  32 40 40 256 128 1 1  32 160 160 32 32 1 1  32 80 80 96 64 1 1  32 40 40 192 128 1 1  32 20 20 384 256 1 1  32 80 80 80 80 1 1 2>&1 | grep -E "GFLOP|res3|res1|ptch|bm 128 dma  bk 16|bm  64 dma  bk 16|bm 128 regs|bm 128 role" | tee $O/res_f32_shape_ab.txt
if [ "${2:-}" = engine ]; then
for res in 7 0 7 0; do
  TRTX_CONV_RES=$res timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_res$res.json 2> $O/bench_res$res.err
  python - $O/bench_res$res.json $res <<'P' | tee -a $O/engine_f32.txt
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line); t = r["tolerance_engine"]
        print(f"TRTX_CONV_RES={sys.argv[2]}: fp32 engine {t['value']:.0f} img/s ({t['ms_per_step']:.3f} ms/step), single context {t['single_context']['ms_per_step']:.3f} ms, conv frac {t['roofline']['frac']:.3f}, conv ms {t['roofline']['conv_ms_per_step']:.3f};  fp16 value {r['value']:.0f}")
P
done
fi
