#!/bin/bash
# Round-2 final evidence (after tactic tuning, compact epilogues, contexts in flight): bench lines for every BASELINE config,
# rocprofv3 kernel stats of the same commands, PMC HBM traffic of the conv kernels, SQ counters of the 64->64 3x3 80x80 layer.
export TMPDIR=/tmp
# tactic timing is on by default; the runs marked "untuned" set TRTX_TUNE=0 (static default launch configurations)
# one tactic cache for every process of this script: the bench runs time the layers once, the rocprofv3 runs then see only the
# real launches (no candidate timing runs inside their kernel statistics)
export TRTX_TACTIC_CACHE=/tmp/trtx_tactics.txt
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/evidence_r02b
mkdir -p $E
TRTX_TUNE_VERBOSE=1 timeout 400 python bench.py --dump-ops $E/ops_c3.json > $E/bench_c3.log 2> $E/bench_c3.err; tail -1 $E/bench_c3.log | cut -c1-200
timeout 300 python bench.py --contexts 1 --no-cpu-baseline --dump-ops $E/ops_c3_1ctx.json > $E/bench_c3_1ctx.log 2>&1; tail -1 $E/bench_c3_1ctx.log | cut -c1-200
TRTX_TUNE=0 timeout 300 python bench.py --contexts 1 --no-cpu-baseline --dump-ops $E/ops_c3_1ctx_untuned.json > $E/bench_c3_1ctx_untuned.log 2>&1; tail -1 $E/bench_c3_1ctx_untuned.log | cut -c1-200
TRTX_TUNE=0 timeout 300 python bench.py --no-cpu-baseline > $E/bench_c3_untuned.log 2>&1; tail -1 $E/bench_c3_untuned.log | cut -c1-200
for cfg in resnet50 retinaface_r50 rcnn_r50c4; do
  timeout 400 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $E/bench_$cfg.log 2>&1; tail -1 $E/bench_$cfg.log | cut -c1-200
done
timeout 400 python bench.py --precision int8 --no-cpu-baseline > $E/bench_c3_int8.log 2>&1; tail -1 $E/bench_c3_int8.log | cut -c1-200
prof() {  # name, bench args
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$1 -o p -- python $R/bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline > $E/prof_$1.log 2>&1)
  python tools/rocprof_summary.py $E/prof_$1 > $E/kernel_stats_$1.txt 2>&1; head -8 $E/kernel_stats_$1.txt | cut -c1-180
  rm -rf $E/prof_$1
}
prof c3 ""
# one context on ONE lane: no kernels overlap, so per-kernel durations are comparable with bench.py's serialized hipEvent profile
TRTX_LANES=1 prof c3_1ctx_lanes1 "--contexts 1"
prof c5 "--config rcnn_r50c4"
prof c2 "--config resnet50"
prof c4 "--config retinaface_r50"
# PMC pass (its own run: --pmc with --kernel-trace only)
OUT=$E/pmc
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT -o c -- python $R/bench.py --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1)
python - <<PY
import csv, glob, collections, json
f = glob.glob("$OUT/**/c_counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    fam = "conv" if ("conv_igemm" in k or "conv_ws" in k) else ("conv_stem" if "conv_stem" in k else ("yolo" if "yolo" in k else "other"))
    per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], fam)
    if key not in seen:
        seen.add(key); n[fam] += 1
lines = ["# rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -- python bench.py --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline (round 2, final build)",
         "# bytes = (2 x RDREQ + WRREQ) x 64 B  (reads doubled: gfx950 counts 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section)"]
res = {}
for fam, d in per.items():
    rd, wr = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_WRREQ_sum", 0.0)
    byts = (2 * rd + wr) * 64
    res[fam] = byts / max(n[fam], 1)
    lines.append(f"{fam:12s} launches {n[fam]:6d}  RDREQ {rd:14.0f}  WRREQ {wr:14.0f}  bytes/launch {byts / max(n[fam], 1):14.0f}")
open("$E/pmc_conv_traffic.txt", "w").write("\n".join(lines) + "\n")
json.dump({"yolov8n": {"bytes_per_launch": res.get("conv"), "source": "profiles/r02_pmc_conv_traffic.txt (separate rocprofv3 --pmc pass over bench.py, fused MFMA conv kernels: conv_igemm* + conv_ws*)"}}, open("$E/pmc_conv_traffic.json", "w"), indent=1)
print("\n".join(lines))
PY
rm -rf $OUT
cp /tmp/trtx_tactics.txt $E/tactic_cache.txt
# SQ counters of one layer, default tactic vs row-reuse kernel
bash tools/pmc_conv2.sh "64 64 3 1 80" sq_default > $E/sq_64x64_3x3_80_default.txt 2>&1
TRTX_TACTIC=64,32,128,1,1,1 bash tools/pmc_conv2.sh "64 64 3 1 80" sq_r3 > $E/sq_64x64_3x3_80_r3.txt 2>&1
rm -rf $R/gpurun_out/pmc2_*
tail -30 $E/sq_64x64_3x3_80_default.txt
