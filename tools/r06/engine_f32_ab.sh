#!/bin/bash
# Round 6: the fp32 (tolerance) engine and the fp16 headline with / without the resident-operand kernels (TRTX_CONV_RES), same box, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r06_ef32}; mkdir -p $O; cd $R
for res in 7 0 7 0; do
  TRTX_CONV_RES=$res timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_res$res.json 2> $O/bench_res$res.err
  python - $O/bench_res$res.json $res <<'P' | tee -a $O/engine_f32.txt
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line); t = r["tolerance_engine"]
        print(f"TRTX_CONV_RES={sys.argv[2]}: fp32 engine {t['value']:.0f} img/s ({t['ms_per_step']:.3f} ms/step), single context {t['single_context']['ms_per_step']:.3f} ms, conv frac {t['roofline']['frac']:.3f}, conv ms {t['roofline']['conv_ms_per_step']:.3f};  fp16 value {r['value']:.0f} single {r['single_context']['ms_per_step']:.3f} ms frac {r['roofline']['frac']:.3f}")
P
done
