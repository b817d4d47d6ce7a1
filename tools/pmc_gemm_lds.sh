cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 6 9; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_l${c}_A -o p -- python3 $R/tools/gemm_one.py 8192 8192 8192 $c > $R/gpurun_out/pmc_l${c}_A.log 2>&1
done
cd $R
python3 tools/pmc_read.py l6 > gpurun_out/s64_l6.log 2>&1
python3 tools/pmc_read.py l9 > gpurun_out/s64_l9.log 2>&1
cat gpurun_out/s64_l6.log gpurun_out/s64_l9.log
rm -rf gpurun_out/pmc_l6_* gpurun_out/pmc_l9_*
