# rocprofv3 counter passes over the heaviest in-step operator shapes (profiles/r3/pmc_ops_r3.md).  usage: bash tools/pmc_ops.sh
# Separate passes: SQ counters | FETCH_SIZE | WRITE_SIZE (TCC slots), each with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while read -r ARGS; do
  [ -z "$ARGS" ] && continue
  i=$((i+1))
  for P in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "rd:FETCH_SIZE" "wr:WRITE_SIZE"; do
    tag=${P%%:*}; ctr=${P#*:}
    rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/pmcops/op${i}_$tag -o p -- python3 $R/tools/op_one.py $ARGS > $R/gpurun_out/pmcops/op${i}_$tag.log 2>&1
  done
  echo "$ARGS" > $R/gpurun_out/pmcops/op${i}.args
  echo "done $i: $ARGS"
done <<'LIST'
geglu 49152 5120 640
geglu 12288 10240 1280
gemm 196608 320 320 res
gemm 196608 960 320
gemm 49152 640 2560 res
gemm 12288 1280 5120 res
conv 48 64 320 320
conv 48 32 640 640
conv 48 8 1280 1280
ffn
ffn_po
rowgemm 960
rowgemm 960 vt
rowgemm 320
attn
LIST
python3 $R/tools/pmc_table.py $R/gpurun_out/pmcops > $R/gpurun_out/pmc_ops_r3d.md
find $R/gpurun_out/pmcops -name "*.db" -delete; find $R/gpurun_out/pmcops -name "*.csv" -size +2M -delete
