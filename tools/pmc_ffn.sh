# rocprofv3 PMC passes over the fused FeedForward kernel.  usage: bash tools/pmc_ffn.sh TAG [dbg]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; DBG=$2
run() { rocprofv3 --kernel-trace --pmc $2 -d $R/gpurun_out/pmc_${TAG}_$1 -o p -- python3 $R/tools/ffn_one.py $DBG > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1; }
run A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run B "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
run C "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16"
