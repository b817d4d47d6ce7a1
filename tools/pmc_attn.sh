# rocprofv3 PMC passes over the spatial attention kernel (hd 40, bank).  usage: bash tools/pmc_attn.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmca_${TAG}_$1 -o p -- python3 $R/tools/attn_one.py > $R/gpurun_out/pmca_${TAG}_$1.log 2>&1; }
run A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
run B "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC"
run C "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES"
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/pmca_${TAG}_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            if "attn" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        for k, v in acc.items():
            print(f"{k:32s} per launch {v / max(n[k],1):16.6g}   launches {n[k]}")
PY
