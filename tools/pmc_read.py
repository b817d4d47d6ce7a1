#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc counters of the gemm kernel per launch from gpurun_out/pmc_<TAG>_*/p_results.db."""
import glob
import sqlite3
import sys

tag = sys.argv[1]
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for db in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/p_results.db")):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    t = lambda k: next(x for x in tabs if k in x)
    q = (f"select s.kernel_name, i.name, sum(e.value) from {t('rocpd_pmc_event')} e join {t('rocpd_info_pmc')} i on e.pmc_id=i.id "
         f"join {t('rocpd_kernel_dispatch')} k on e.event_id=k.event_id join {t('rocpd_info_kernel_symbol')} s on k.kernel_id=s.id "
         f"group by s.kernel_name, i.name")
    for name, ctr, val in con.execute(q):
        if "gemm" in name or "ff_fused" in name:
            print(f"{tag:10s} {ctr:45s} {val / nl:14.4g}")
