# rocprofv3 PMC passes over one GEMM shape.  usage: bash tools/pmc_gemm.sh TAG "M N K cfg [res]"   (MMGT_ALT_LIB optional)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; ARGS=$2
run() { rocprofv3 --kernel-trace --pmc $2 -d $R/gpurun_out/pmc_${TAG}_$1 -o p -- python3 $R/tools/gemm_one.py $ARGS > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1; }
run A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES"
run C "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
run E "GRBM_GUI_ACTIVE TCC_BUSY_sum TA_TA_BUSY_sum TCC_TAG_STALL_sum"
