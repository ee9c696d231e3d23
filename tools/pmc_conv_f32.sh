# usage (on the GPU box): bash tools/pmc_conv_f32.sh "<N H W Cin Cout k stride>" "<bn,bm,ws>" tag  -> SQ / cache counters of the fp32 MFMA conv kernel (one --pmc pass per set)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS="$1"; ONLY="$2"; TAG="$3"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  OUT=$R/gpurun_out/pmcf32_${TAG}_$i
  (cd /tmp && timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT -o c -- python $R/tools/conv_f32_shape_ab.py --only $ONLY $ARGS > $OUT.log 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmcf32_${TAG}_*/**/c_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("$TAG", c, round(sum(v) / len(v)), "(%d launches)" % len(v))
PY
