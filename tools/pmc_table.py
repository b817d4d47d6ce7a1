#!/usr/bin/env python3
"""Condense the passes of tools/pmc_ops.sh into one markdown table: per operator shape the kernel's average duration, the matrix
pipe's busy share (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles), shader cycles = GRBM_GUI_ACTIVE / 8 XCDs), the wave
states, and the HBM-side bytes (FETCH_SIZE x 2: the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE) per launch."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]


def counters(db):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    t = lambda k: next(x for x in tabs if k in x)
    q = (f"select s.kernel_name, i.name, sum(e.value), count(distinct k.id) from {t('rocpd_pmc_event')} e join {t('rocpd_info_pmc')} i on e.pmc_id=i.id "
         f"join {t('rocpd_kernel_dispatch')} k on e.event_id=k.event_id join {t('rocpd_info_kernel_symbol')} s on k.kernel_id=s.id "
         f"group by s.kernel_name, i.name")
    out = {}
    for name, ctr, val, n in con.execute(q):
        out.setdefault(name, {})[ctr] = (val, n)
    cols = [r[1] for r in con.execute(f"pragma table_info({t('rocpd_kernel_dispatch')})")]
    cs, ce = next(c for c in cols if c.startswith("start")), next(c for c in cols if c.startswith("end"))
    q2 = (f"select s.kernel_name, avg(k.{ce} - k.{cs}), count(*) from {t('rocpd_kernel_dispatch')} k join {t('rocpd_info_kernel_symbol')} s "
          f"on k.kernel_id=s.id group by s.kernel_name")
    dur = {name: (d, n) for name, d, n in con.execute(q2)}
    return out, dur


def main_kernel(ctrs, dur):
    # the operator's own kernel = the one with the largest total time among mmgt kernels
    own = {k: v for k, v in dur.items() if any(s in k for s in ("gemm", "attn", "ff_fused", "rconv"))}
    return max(own, key=lambda k: own[k][0] * own[k][1])


print("| operator shape | kernel | us / launch | matrix pipe busy | wave cycles: active / issue-stalled / waiting | HBM read GB | HBM write GB | GB/s |")
print("|---|---|---|---|---|---|---|---|")
for argf in sorted(glob.glob(os.path.join(root, "op*.args")), key=lambda p: int(os.path.basename(p)[2:-5])):
    tag = argf[:-5]
    args = open(argf).read().strip()
    sq, dur = counters(glob.glob(tag + "_sq/**/*.db", recursive=True)[0])
    rd, _ = counters(glob.glob(tag + "_rd/**/*.db", recursive=True)[0])
    wr, _ = counters(glob.glob(tag + "_wr/**/*.db", recursive=True)[0])
    k = main_kernel(sq, dur)
    c = {n: v[0] / v[1] for n, v in sq[k].items()}
    us = dur[k][0] / 1e3
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
    wc = c["SQ_WAVE_CYCLES"]
    fetch = rd[k]["FETCH_SIZE"][0] / rd[k]["FETCH_SIZE"][1] * 1024 * 2 / 1e9        # KiB -> bytes, x2 (gfx950)
    write = wr[k]["WRITE_SIZE"][0] / wr[k]["WRITE_SIZE"][1] * 1024 / 1e9
    short = k.split("(")[0].replace("(anonymous namespace)::", "")[:60]
    print(f"| {args} | `{short}` | {us:.1f} | {100 * busy:.1f} % | {100 * c['SQ_ACTIVE_INST_ANY'] / wc:.0f} / {100 * c['SQ_WAIT_INST_ANY'] / wc:.0f} / "
          f"{100 * c['SQ_WAIT_ANY'] / wc:.0f} % | {fetch:.3f} | {write:.3f} | {(fetch + write) / us * 1e6:.0f} |")
