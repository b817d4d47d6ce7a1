#!/usr/bin/env python3
"""Static issue-cost accounting of a kernel's MFMA gaps from its ISA listing, with the measured costs of MI355X_MICROARCH.md
('vector-instruction ISSUE cost' row): transcendental 8 cycles, other VALU / v_cvt_pk / SALU / s_nop 0 4, packed fp32 VALU 8, an MFMA holds
the SIMD's vector issue for 8 of its 32 (32x32x16) or 16 (16x16x32) pipe cycles; a gap runs max(pipe, 8 + sum of its fillers' costs).

    hipcc -O3 ... -S --cuda-device-only kernel.hip -o k.s
    python tools/isa_gaps.py k.s --start .LBB0_56 [--take .LBB0_58 ...] [--stop .LBB0_45]

walks the listing from label --start in program order, follows a branch when its target is listed in --take (the common path of a rare
branch), stops at --stop or at the first backward branch, and prints one line per MFMA gap plus the totals of ONE wave's stream.  It is a
model of ONE wave alone on its SIMD; what two waves per SIMD do to each other is what the in-kernel stamps (tools/trace_attn64.py) measure."""
import argparse
import re

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def cost(op, args):
    if op.startswith("v_mfma"):
        return "mfma", 8
    if op.startswith(TRANS):
        return "trans", 8
    if op.startswith("v_pk_") and "f32" in op:
        return "valu", 8
    if op.startswith("v_"):
        return "valu", 4
    if op == "s_nop":
        return "nop", max(4, int(args.split()[0]) + 1)
    if op.startswith("ds_"):
        return "lds", 2                 # (<= 3 cycles per gap for two ds_read_b128: MI355X_MICROARCH.md, LDS section)
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem", 4
    if op.startswith("s_waitcnt") or op.startswith("s_barrier"):
        return "wait", 0                # (what they wait FOR is not an issue cost; the stamps measure it)
    if op.startswith("s_"):
        return "salu", 4
    return "other", 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing")
    ap.add_argument("--start", required=True)
    ap.add_argument("--take", action="append", default=[])
    ap.add_argument("--stop", default=None)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    lines = open(a.listing).read().split("\n")
    label_at = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    i = label_at[a.start]
    seen_labels = {a.start}
    gaps, cur = [], dict(valu=0, trans=0, nop=0, lds=0, vmem=0, salu=0, other=0, wait=0, cost=0)
    pipe_of_prev = None
    totals = dict(mfma=0, pipe=0, issue=0, pred=0)
    counts = dict(valu=0, trans=0, nop=0, lds=0, vmem=0, salu=0, other=0, wait=0)

    def close(pipe):
        nonlocal cur
        if pipe is not None:
            pred = max(pipe, 8 + cur["cost"])
            gaps.append((pipe, dict(cur), pred))
            totals["pred"] += pred
        else:                               # instructions in front of the first MFMA of the walk
            totals["pred"] += cur["cost"]
        cur = dict(valu=0, trans=0, nop=0, lds=0, vmem=0, salu=0, other=0, wait=0, cost=0)

    while i < len(lines):
        l = lines[i].strip()
        i += 1
        if not l or l.startswith(";") or l.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                if m.group(1) == a.stop:
                    break
                seen_labels.add(m.group(1))
            continue
        op, _, args = l.partition(" ")
        op = op.strip()
        args = args.split(";")[0].strip()
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = args.split()[-1]
            if tgt in a.take or op == "s_branch":
                if tgt in seen_labels and tgt != a.start:
                    break
                if tgt == a.start or tgt == a.stop:
                    break
                i = label_at[tgt]
                seen_labels.add(tgt)
            cur["salu"] += 1
            cur["cost"] += 4
            totals["issue"] += 4
            continue
        kind, c = cost(op, args)
        totals["issue"] += c
        if kind == "mfma":
            close(pipe_of_prev)
            pipe_of_prev = 32 if "32x32" in op else 16
            totals["mfma"] += 1
            totals["pipe"] += pipe_of_prev
        else:
            cur[kind] += 1
            cur["cost"] += c
            counts[kind] += 1
    close(pipe_of_prev)
    if not a.quiet:
        print("gap  pipe  valu trans pk/nop lds vmem salu | filler issue cost | predicted cycles = max(pipe, 8 + cost)")
        for n, (pipe, g, pred) in enumerate(gaps):
            print(f"{n:3d}  {pipe:4d}  {g['valu']:4d} {g['trans']:5d} {g['nop']:6d} {g['lds']:3d} {g['vmem']:4d} {g['salu']:4d} | {g['cost']:17d} | {pred:6d}")
    print(f"\n{totals['mfma']} MFMAs: matrix pipe {totals['pipe']} cycles; issue costs of the whole stream {totals['issue']} cycles "
          f"(VALU {counts['valu']}, transcendental {counts['trans']}, LDS {counts['lds']}, VMEM {counts['vmem']}, SALU {counts['salu']}, s_nop {counts['nop']})")
    print(f"model of ONE wave alone on its SIMD: sum over gaps of max(pipe, 8 + fillers) = {totals['pred']} cycles per iteration")
    print(f"two such waves per SIMD: matrix pipe {2 * totals['pipe']}, issue port {2 * totals['issue']} -> floor max = {max(2 * totals['pipe'], 2 * totals['issue'])} "
          f"cycles per pair of iterations if the waves' streams interleave perfectly")


if __name__ == "__main__":
    main()
