#!/usr/bin/env python3
"""Condense the rocprofv3 CSV outputs of tools/profile_bench.sh into the small files kept under profiles/:
  <out>/bench_steps3_kernel_stats_<tag>.csv   rocprofv3's own per-kernel stats (copied)
  <out>/bench_steps3_summary_<tag>.md         per-kernel launches / ms per step
  <out>/pmc_traffic_<tag>.json                HBM-side read / write bytes per step and per kernel
  <out>/pmc_step_mfma_<tag>.json              matrix-pipe busy fraction of the step and of every kernel (SQ_VALU_MFMA_BUSY_CYCLES)
usage: python3 tools/profile_summarize.py <dir with prof_<tag>_{trace,fetch,write}> <tag> <out dir>"""
import collections
import csv
import json
import os
import re
import shutil
import sys

src, tag, out = sys.argv[1:4]
os.makedirs(out, exist_ok=True)
csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


# ---- kernel time per step
tdir = os.path.join(src, f"prof_{tag}_trace")
shutil.copy(os.path.join(tdir, "p_kernel_stats.csv"), os.path.join(out, f"bench_steps3_kernel_stats_{tag}.csv"))
steps_total = 4                                    # 1 warm-up + 3 timed
agg = collections.OrderedDict()
with open(os.path.join(tdir, "p_kernel_trace.csv")) as f:
    for r in csv.DictReader(f):
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
is_own = lambda k: "_kernel" in k and "at::" not in k and "native" not in k and "rocprim" not in k
own = {k: v for k, v in agg.items() if is_own(k)}
rows = sorted(own.items(), key=lambda kv: -kv[1][1])
bench_line = ""
log = os.path.join(src, f"prof_{tag}_trace.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith('{"metric"'):
            bench_line = line.strip()
with open(os.path.join(out, f"bench_steps3_summary_{tag}.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-calib  (build {tag})\n\n")
    f.write(f"{steps_total} denoise steps (1 warm-up + 3 timed) at 512x512x24 bf16 on one MI355X; per-step = total / {steps_total}. "
            f"Raw stats: bench_steps3_kernel_stats_{tag}.csv; HBM-side traffic from PMC: pmc_traffic_{tag}.json.\n\n")
    f.write("| kernel | launches/step | ms/step | avg us |\n|---|---|---|---|\n")
    tot = 0.0
    for k, (n, ns) in rows:
        f.write(f"| `{k}` | {n / steps_total:.1f} | {ns / steps_total / 1e6:.2f} | {ns / n / 1e3:.1f} |\n")
        tot += ns / steps_total / 1e6
    other = sum(v[1] for k, v in agg.items() if k not in own) / steps_total / 1e6
    f.write(f"\nTotal own-kernel time per step: {tot:.1f} ms (torch-side kernels, mostly setup outside the timed steps: {other:.2f} ms).\n")
    if bench_line:
        b = json.loads(bench_line)
        f.write(f"\nbench.py line of the same (profiled) run: {b['ms_per_step']:.1f} ms/step, {b['value']:.2f} steps/s, "
                f"roofline.achieved {b['roofline']['achieved']:.0f} TFLOP/s = {100 * b['roofline']['frac']:.1f}% of 2.5 PF.\n")

# ---- PMC traffic
def pmc(kind, counter):
    d = os.path.join(src, f"prof_{tag}_{kind}", "p_counter_collection.csv")
    per = collections.defaultdict(lambda: [0.0, 0])
    with open(d) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            per[k][0] += float(r["Counter_Value"])
            per[k][1] += 1
    return per


steps_pmc = 4                                      # 1 warm-up + 3 timed: the same command as the kernel-trace pass
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
perk = {}
for k in set(fetch) | set(write):
    if not is_own(k):
        continue
    perk[k] = {"read": 2 * fetch[k][0] * 1024 / steps_pmc / 1e9, "write": write[k][0] * 1024 / steps_pmc / 1e9,
               "launches_per_step": max(fetch[k][1], write[k][1]) / steps_pmc}
perk = dict(sorted(perk.items(), key=lambda kv: -(kv[1]["read"] + kv[1]["write"])))
rd, wr = sum(v["read"] for v in perk.values()), sum(v["write"] for v in perk.values())
json.dump({
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-calib",
    "steps_profiled": steps_pmc,
    "correction": "gfx950: FETCH_SIZE reports exactly 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> read bytes = "
                  "2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16-byte stores.",
    "note": "counts L2 -> fabric requests, i.e. Infinity-Cache hits are included (not pure HBM)",
    "read_GB_per_step": rd, "write_GB_per_step": wr, "per_kernel_GB_per_step": perk, "total_GB_per_step": rd + wr},
    open(os.path.join(out, f"pmc_traffic_{tag}.json"), "w"), indent=1)
# ---- matrix-pipe busy: SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles summed over the chip's 1024 SIMDs; a kernel's cycles = GRBM_GUI_ACTIVE / 8
# (rocprofv3 sums the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back)
mdir = os.path.join(src, f"prof_{tag}_mfma", "p_counter_collection.csv")
if os.path.exists(mdir):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    with open(mdir) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if not is_own(k):
                continue
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
    NSIMD = 1024
    rowsm = {}
    for k, c in per.items():
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        rowsm[k] = {"mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (NSIMD * cyc) if cyc else 0.0, "kernel_cycles_per_step": cyc / steps_pmc,
                    "launches_per_step": cnt[k] / steps_pmc}
    tot_busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in per.values())
    tot_cyc = sum(c["GRBM_GUI_ACTIVE"] / 8 for c in per.values())
    rowsm = dict(sorted(rowsm.items(), key=lambda kv: -kv[1]["kernel_cycles_per_step"]))
    json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 "
                          "--warmup 1 --no-cpu-baseline --no-extras --no-calib",
               "steps_profiled": steps_pmc, "mfma_busy": tot_busy / (NSIMD * tot_cyc),
               "definition": "sum over the step's own kernels of SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum of kernel cycles), kernel cycles = "
                             "GRBM_GUI_ACTIVE / 8; a profiled pass holds a lower clock than an un-profiled run, cycles are what is compared",
               "per_kernel": rowsm}, open(os.path.join(out, f"pmc_step_mfma_{tag}.json"), "w"), indent=1)
    print("matrix pipe busy over the step:", round(tot_busy / (NSIMD * tot_cyc), 4))
print(open(os.path.join(out, f"bench_steps3_summary_{tag}.md")).read()[-900:])
print("traffic GB/step: read", round(rd, 1), "write", round(wr, 1))
