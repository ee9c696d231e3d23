#!/bin/bash
# Round-4 evidence, all of it from ONE build: run on an MI355X box through gpurun (`gpurun -- bash tools/evidence_r04.sh [part]`), results under
# gpurun_out/evidence_r04/, copied into profiles/r03_* by hand afterwards (profiles/README.md says which file came from which part).
#   part "bench"  : bench.py lines - C3 with the driver's flags (three times), one context, TRTX_TUNE=0, int8, C2 / C4 / C5 - and the layer table
#   part "prof"   : rocprofv3 --kernel-trace --stats of the same commands (three contexts; one context on one lane; C2 / C4 / C5), the PMC
#                   pass for HBM traffic (its own run: --pmc with --kernel-trace only)
#   part "suite"  : the full `pytest -m gpu` (writes gpurun_out/parity_metrics.jsonl; run on several boxes for the parity record)
export TMPDIR=/tmp
export TRTX_TACTIC_CACHE=/tmp/trtx_tactics.txt   # one set of tactic timings for every process of this script: the rocprofv3 runs see real launches only
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/evidence_r04
mkdir -p $E
cd $R
PART=${1:-all}
if [ $PART = bench ] || [ $PART = all ]; then
  for i in 1 2 3; do
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $E/bench_c3_driverflags_$i.log 2> $E/bench_c3_driverflags_$i.err
  done
  timeout 300 python bench.py --steps 20 --warmup 5 --contexts 1 --no-cpu-baseline --dump-ops $E/ops_c3_1ctx.json > $E/bench_c3_1ctx.log 2>/dev/null
  TRTX_TUNE=0 TRTX_TACTIC_CACHE= timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $E/bench_c3_untuned.log 2>/dev/null
  TRTX_FOLD_UPSAMPLE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $E/bench_c3_nofold.log 2>/dev/null
  timeout 500 python bench.py --steps 20 --warmup 5 --precision int8 > $E/bench_c3_int8.log 2>/dev/null
  for cfg in resnet50 retinaface_r50 rcnn_r50c4; do
    timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 > $E/bench_$cfg.log 2>/dev/null
  done
  python tools/show_bench.py $E/bench_*.log
  timeout 300 python tools/layer_table.py $E/layer_table.json > $E/layer_table.txt 2>&1; tail -5 $E/layer_table.txt
fi
if [ $PART = prof ] || [ $PART = all ]; then
  timeout 300 python bench.py --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > /dev/null 2>&1            # fills the tactic cache (3 contexts + 1 context engines)
  prof() {  # name, bench args
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$1 -o p -- python $R/bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline > $E/prof_$1.log 2>&1)
    { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline   (round 4, final build; TRTX_TACTIC_CACHE set: no tactic-timing launches inside)"; python tools/rocprof_summary.py $E/prof_$1; } > $E/kernel_stats_$1.txt 2>&1
    head -9 $E/kernel_stats_$1.txt | cut -c1-170
    rm -rf $E/prof_$1
  }
  prof c3 ""
  TRTX_LANES=1 prof c3_1ctx_lanes1 "--contexts 1"     # no kernels overlap: per-kernel durations comparable with bench.py's serialized profile passes
  for cfg in resnet50 retinaface_r50 rcnn_r50c4; do timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline > /dev/null 2>&1; done
  prof c2 "--config resnet50"
  prof c4 "--config retinaface_r50"
  prof c5 "--config rcnn_r50c4"
  # single context on one lane: no kernels overlap, the durations are the kernels' own
  TRTX_LANES=1 prof c2_1ctx_lanes1 "--config resnet50 --contexts 1"
  TRTX_LANES=1 prof c4_1ctx_lanes1 "--config retinaface_r50 --contexts 1"
  TRTX_LANES=1 prof c5_1ctx_lanes1 "--config rcnn_r50c4 --contexts 1"
fi
if [ $PART = c3prof ]; then   # the C3 profiles and PMC pass alone (after the last small kernel changes of the round)
  timeout 300 python bench.py --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > /dev/null 2>&1
  prof() {
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$1 -o p -- python $R/bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline > $E/prof_$1.log 2>&1)
    { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline   (round 4, final build; TRTX_TACTIC_CACHE set: no tactic-timing launches inside)"; python tools/rocprof_summary.py $E/prof_$1; } > $E/kernel_stats_$1.txt 2>&1
    head -9 $E/kernel_stats_$1.txt | cut -c1-170
    rm -rf $E/prof_$1
  }
  prof c3 ""
  TRTX_LANES=1 prof c3_1ctx_lanes1 "--contexts 1"
  OUT=$E/pmc_yolov8n
  (cd /tmp && timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT -o c -- python $R/bench.py --config yolov8n --contexts 1 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline > $OUT.log 2>&1)
  python - <<PY
import csv, glob, collections
fs = glob.glob("$E/pmc_yolov8n/**/c_counter_collection.csv", recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    fam = "conv" if ("conv_igemm" in k or "conv_ws" in k or "conv_gemm256" in k) else ("conv_stem" if "conv_stem" in k else ("yolo" if "yolo" in k else "other"))
    per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], fam)
    if key not in seen:
        seen.add(key); n[fam] += 1
for fam, d in per.items():
    rd, wr = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_WRREQ_sum", 0.0)
    print(f"yolov8n {fam:12s} launches {n[fam]:6d}  RDREQ {rd:14.0f}  WRREQ {wr:14.0f}  bytes/launch {(2 * rd + wr) * 64 / max(n[fam], 1):14.0f}")
PY
  rm -rf $E/pmc_yolov8n/
fi
if [ $PART = c5 ]; then   # C5 again after the last kernel change of the round (conv_gemm256 on the 3x3, shortcut prefetch): bench line, both profiles, PMC
  timeout 600 python bench.py --config rcnn_r50c4 --steps 20 --warmup 5 > $E/bench_rcnn_r50c4.log 2>/dev/null
  prof() {
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$1 -o p -- python $R/bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline > $E/prof_$1.log 2>&1)
    { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline   (round 4, final build; TRTX_TACTIC_CACHE set: no tactic-timing launches inside)"; python tools/rocprof_summary.py $E/prof_$1; } > $E/kernel_stats_$1.txt 2>&1
    head -9 $E/kernel_stats_$1.txt | cut -c1-170
    rm -rf $E/prof_$1
  }
  prof c5 "--config rcnn_r50c4"
  TRTX_LANES=1 prof c5_1ctx_lanes1 "--config rcnn_r50c4 --contexts 1"
  OUT=$E/pmc_rcnn_r50c4
  (cd /tmp && timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT -o c -- python $R/bench.py --config rcnn_r50c4 --contexts 1 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline > $OUT.log 2>&1)
  python - <<PY
import csv, glob, collections
fs = glob.glob("$E/pmc_rcnn_r50c4/**/c_counter_collection.csv", recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    fam = "conv" if ("conv_igemm" in k or "conv_ws" in k or "conv_gemm256" in k) else ("conv_stem" if "conv_stem" in k else "other")
    per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], fam)
    if key not in seen:
        seen.add(key); n[fam] += 1
for fam, d in per.items():
    rd, wr = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_WRREQ_sum", 0.0)
    print(f"rcnn_r50c4 {fam:12s} launches {n[fam]:6d}  RDREQ {rd:14.0f}  WRREQ {wr:14.0f}  bytes/launch {(2 * rd + wr) * 64 / max(n[fam], 1):14.0f}")
PY
  rm -rf $E/pmc_rcnn_r50c4/
fi
if [ $PART = prof1 ]; then   # only the single-context kernel statistics of C2 / C4 / C5 (a short call early in the round)
  prof() {
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$1 -o p -- python $R/bench.py $2 --steps 10 --warmup 3 --repeats 2 --no-cpu-baseline > $E/prof_$1.log 2>&1)
    { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $2 --steps 10 --warmup 3 --repeats 2 --no-cpu-baseline   (round 4; TRTX_LANES=1, one context: no kernels overlap)"; python tools/rocprof_summary.py $E/prof_$1; } > $E/kernel_stats_$1.txt 2>&1
    head -14 $E/kernel_stats_$1.txt | cut -c1-170
    rm -rf $E/prof_$1
  }
  for cfg in resnet50 retinaface_r50 rcnn_r50c4; do timeout 300 python bench.py --config $cfg --contexts 1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > /dev/null 2>&1; done
  TRTX_LANES=1 prof c2_1ctx_lanes1 "--config resnet50 --contexts 1"
  TRTX_LANES=1 prof c4_1ctx_lanes1 "--config retinaface_r50 --contexts 1"
  TRTX_LANES=1 prof c5_1ctx_lanes1 "--config rcnn_r50c4 --contexts 1"
fi
if [ $PART = pmc ] || [ $PART = prof ] || [ $PART = all ]; then
  # HBM traffic of the conv launches from the L2 fabric counters, one pass per configuration (its own run: --pmc with --kernel-trace only)
  for cfg in yolov8n resnet50 retinaface_r50 rcnn_r50c4; do
    OUT=$E/pmc_$cfg
    timeout 300 python bench.py --config $cfg --contexts 1 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > /dev/null 2>&1      # tactic cache
    (cd /tmp && timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT -o c -- python $R/bench.py --config $cfg --contexts 1 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline > $OUT.log 2>&1)
  done
  python - <<PY
import csv, glob, collections, json
lines = ["# rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -- python bench.py --config X --contexts 1 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline (round 4, final build)",
         "# bytes = (2 x RDREQ + WRREQ) x 64 B  (reads doubled: gfx950 counts 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section); per launch, mean over the launches of the run"]
out = {}
for cfg in ("yolov8n", "resnet50", "retinaface_r50", "rcnn_r50c4"):
    fs = glob.glob("$E/pmc_%s/**/c_counter_collection.csv" % cfg, recursive=True)
    if not fs:
        lines.append(f"{cfg}: no counter file")
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        fam = "conv" if ("conv_igemm" in k or "conv_ws" in k or "conv_gemm256" in k) else ("conv_stem" if "conv_stem" in k else ("yolo" if "yolo" in k else "other"))
        per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], fam)
        if key not in seen:
            seen.add(key); n[fam] += 1
    lines.append(f"== {cfg}")
    for fam, d in per.items():
        rd, wr = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_WRREQ_sum", 0.0)
        byts = (2 * rd + wr) * 64
        lines.append(f"{fam:12s} launches {n[fam]:6d}  RDREQ {rd:14.0f}  WRREQ {wr:14.0f}  bytes/launch {byts / max(n[fam], 1):14.0f}")
        if fam == "conv":
            out[cfg] = {"bytes_per_launch": byts / max(n[fam], 1), "source": "profiles/r04_pmc_conv_traffic.txt (separate rocprofv3 --pmc pass over bench.py --config %s, fused MFMA conv kernels: conv_igemm* + conv_ws*)" % cfg}
open("$E/pmc_conv_traffic.txt", "w").write("\n".join(lines) + "\n")
json.dump(out, open("$E/pmc_conv_traffic.json", "w"), indent=1)
print("\n".join(lines))
PY
  rm -rf $E/pmc_*/
fi
if [ $PART = suite ] || [ $PART = all ]; then
  unset TRTX_TACTIC_CACHE
  rm -f gpurun_out/parity_metrics.jsonl
  for run in 1 2; do    # two full runs: the parity record needs several (tests/test_parity_bounds.py)
    timeout 1800 python -m pytest tests -m gpu -q > $E/gpu_suite_run$run.log 2>&1
    tail -4 $E/gpu_suite_run$run.log
  done
  cp gpurun_out/parity_metrics.jsonl $E/parity_metrics.jsonl
fi
