# HBM traffic of the conv kernels over bench.py steps from the L2 fabric counters (own pass: --pmc + --kernel-trace only).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
rm -rf $OUT
(cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1)
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/c_counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    fam = "conv_igemm" if "conv_igemm" in k else ("conv_stem" if "conv_stem" in k else ("yolo" if "yolo" in k else "other"))
    per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], fam)
    if key not in seen:
        seen.add(key)
        n[fam] += 1
lines = ["# rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum over bench.py --steps 3 --warmup 1 (3 steps + 1 warm-up + 5 per-op profile passes)",
         "# bytes = (2 x RDREQ + WRREQ) x 64 B  (reads doubled: gfx950 counts 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section)"]
for fam, d in per.items():
    rd, wr = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_WRREQ_sum", 0.0)
    byts = (2 * rd + wr) * 64
    lines.append(f"{fam:12s} launches {n[fam]:6d}  RDREQ {rd:14.0f}  WRREQ {wr:14.0f}  bytes/launch {byts / max(n[fam], 1):14.0f}")
open("$GRAFT_REPO_ROOT/gpurun_out/pmc_bench_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
