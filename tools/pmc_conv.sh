# usage: bash tools/pmc_conv.sh "<cin cout k s hin>" tag   -> L2 hit/miss + fabric read requests of the conv kernel
export TMPDIR=/tmp
ARGS="$1"; TAG="$2"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT -o c -- python $GRAFT_REPO_ROOT/tools/conv_one.py $ARGS > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/c_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "conv_igemm" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("$TAG", k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
