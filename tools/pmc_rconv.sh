# rocprofv3 SQ counter pass over the fused GroupNorm + SiLU + conv launch and the gemm16 conv of the same shape.  usage: bash tools/pmc_rconv.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcrc
i=0
while read -r ARGS; do
  [ -z "$ARGS" ] && continue
  i=$((i+1))
  for P in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "sq2:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "rd:FETCH_SIZE" "wr:WRITE_SIZE"; do
    tag=${P%%:*}; ctr=${P#*:}
    rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/pmcrc/op${i}_$tag -o p -- python3 $R/tools/op_one.py $ARGS > $R/gpurun_out/pmcrc/op${i}_$tag.log 2>&1
  done
  echo "$ARGS" > $R/gpurun_out/pmcrc/op${i}.args
  echo "done $i: $ARGS"
done <<'LIST'
rconv 48 64 320 320
rconv 48 64 320 320 res
conv 48 64 320 320
rconv 48 32 640 640
conv 48 32 640 640
rconv 48 16 1280 1280
conv 48 16 1280 1280
LIST
python3 $R/tools/pmc_table.py $R/gpurun_out/pmcrc > $R/gpurun_out/pmc_rconv_r6.md
python3 - <<'PY' >> $R/gpurun_out/pmc_rconv_r6.md
import glob, os, sqlite3
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmcrc"
print("\nInstruction counters per launch (SQ_INSTS_* are per wave-instruction, summed over the chip):\n")
for d in sorted(glob.glob(root + "/op*_sq2")):
    dbs = glob.glob(d + "/**/*.db", recursive=True)
    if not dbs: continue
    con = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    t = lambda k: next(x for x in tabs if k in x)
    q = (f"select s.kernel_name, i.name, sum(e.value), count(distinct k.id) from {t('rocpd_pmc_event')} e join {t('rocpd_info_pmc')} i on e.pmc_id=i.id "
         f"join {t('rocpd_kernel_dispatch')} k on e.event_id=k.event_id join {t('rocpd_info_kernel_symbol')} s on k.kernel_id=s.id group by s.kernel_name, i.name")
    rows = {}
    for name, ctr, val, n in con.execute(q):
        if any(x in name for x in ("rconv", "gemm16")): rows.setdefault(name[:60], {})[ctr] = val / n
    print(open(d.replace("_sq2", "") + ".args").read().strip(), {k: {c: f"{v:.3g}" for c, v in r.items()} for k, r in rows.items()})
PY
find $R/gpurun_out/pmcrc -name "*.db" -delete; find $R/gpurun_out/pmcrc -name "*.csv" -size +2M -delete
