# usage: bash tools/pmc_conv2.sh "<cin cout k s hin>" tag  -> SQ-level counters of the conv kernel (several passes)
export TMPDIR=/tmp
ARGS="$1"; TAG="$2"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES"; do
  i=$((i+1))
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2_${TAG}_$i
  (cd /tmp && rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT -o c -- python $GRAFT_REPO_ROOT/tools/conv_one.py $ARGS > $OUT.log 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc2_${TAG}_*/**/c_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"] or "conv_ws" in r["Kernel_Name"] or "conv_gemm256" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("$TAG", c, round(sum(v) / len(v)))
PY
