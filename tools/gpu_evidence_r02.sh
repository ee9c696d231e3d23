#!/bin/bash
# Round-2 evidence: bench lines for every BASELINE config, rocprofv3 kernel stats of the same commands, PMC HBM traffic of the conv kernels.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/evidence_r02
mkdir -p $E
timeout 600 python bench.py > $E/bench_c3.log 2>&1; tail -1 $E/bench_c3.log | cut -c1-300
for cfg in resnet50 retinaface_r50 rcnn_r50c4; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $E/bench_$cfg.log 2>&1; tail -1 $E/bench_$cfg.log | cut -c1-300
done
timeout 600 python bench.py --precision int8 --no-cpu-baseline > $E/bench_c3_int8.log 2>&1; tail -1 $E/bench_c3_int8.log | cut -c1-300
prof() {  # name, bench args
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$1 -o p -- python $R/bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline > $E/prof_$1.log 2>&1)
  python tools/rocprof_summary.py $E/prof_$1 > $E/kernel_stats_$1.txt 2>&1; head -12 $E/kernel_stats_$1.txt | cut -c1-200
  rm -rf $E/prof_$1
}
prof c3 ""
# the same step on ONE lane: no kernels overlap, so per-kernel durations are comparable with bench.py's serialized hipEvent profile
TRTX_LANES=1 prof c3_lanes1 ""
prof c2 "--config resnet50"
prof c4 "--config retinaface_r50"
prof c5 "--config rcnn_r50c4"
prof c3_int8 "--precision int8"
# PMC pass (its own run: --pmc with --kernel-trace only)
OUT=$E/pmc
(cd /tmp && timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT -o c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1)
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
lines = ["# rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline (round 2)",
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
python tools/layer_table.py $E/layers.json > $E/layers.txt 2>&1; tail -1 $E/layers.txt
