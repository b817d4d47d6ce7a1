# Round profile of the bench command (run on the GPU box through gpurun): kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in separate PMC passes, then the matrix-pipe counters of the whole step.  usage: bash tools/profile_bench.sh TAG      -> gpurun_out/prof_TAG_{trace,fetch,write}/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-v}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-calib > $R/gpurun_out/prof_${TAG}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-calib > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-calib > $R/gpurun_out/prof_${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_mfma -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-calib > $R/gpurun_out/prof_${TAG}_mfma.log 2>&1
rm -f $R/gpurun_out/prof_${TAG}_*/p_kernel_trace.csv.bak


# keep only the condensed files (gpurun merges at most 64 MiB back)
python3 $R/tools/profile_summarize.py $R/gpurun_out ${TAG} $R/gpurun_out/profile_${TAG}
rm -rf $R/gpurun_out/prof_${TAG}_trace $R/gpurun_out/prof_${TAG}_fetch $R/gpurun_out/prof_${TAG}_write $R/gpurun_out/prof_${TAG}_mfma
