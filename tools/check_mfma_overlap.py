#!/usr/bin/env python3
"""Scan the device code of the built objects (mmgt_amd/csrc/build/*.o) for two instruction patterns hipcc 7.2 emits for gfx950 and the hardware
does not execute as written (both found in csrc/gnconv.hip, round 5):

(1) a 12- / 16-byte VMEM store whose data registers are written by a VALU instruction within the next two wait states.  The hazard recognizer
    inserts the wait states only when the store's soffset is NOT a register (SIInstrInfo / GCNHazardRecognizer: "this hazard only exists if the
    instruction is not using a register in the soffset field"); on MI355X the store then sends the overwritten value for the last lanes of each
    16-lane row (wrong outputs in lanes 12 .. 15, deterministic once the overwrite is the very next instruction).
(2) (reported, not an error) MFMA instructions whose destination registers PARTIALLY overlap their accumulator input (vDst != SrcC but sharing
    registers): hipcc emits them where it rotates accumulator tiles; tools/micro/mfma_overlap.hip shows MI355X computes them correctly.  Exit status 1 and a listing if a store hazard is found.

    python tools/check_mfma_overlap.py [objects ...]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r"(v_s?mfma\w*)\s+([av])\[(\d+):(\d+)\],\s*[av]\[\d+:\d+\],\s*[av]\[\d+:\d+\],\s*([av])\[(\d+):(\d+)\]")


STORE = re.compile(r"((?:buffer|global|flat|scratch)_store_dwordx[34])\s+(?:v\d+,\s*|v\[\d+:\d+\],\s*)?v\[(\d+):(\d+)\]")
VDEF = re.compile(r"^\s*(v_\w+)\s+v(?:\[(\d+):(\d+)\]|(\d+))(?:,\s*v(?:\[(\d+):(\d+)\]|(\d+)))?")


def store_hazards(lines, count=None):
    """(function, store, overwriting instruction) for stores of more than 8 bytes whose data is rewritten within two wait states; count[0] += the
    wide stores parsed"""
    out, func = [], "?"
    for k, line in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func = m.group(1)
            continue
        text = line.split("//")[0]
        m = STORE.search(text)
        if not m:
            continue
        # buffer stores: (vdata, vaddr ...): vdata is the FIRST register operand
        ops = text.split(None, 1)[1] if len(text.split(None, 1)) > 1 else ""
        md = re.match(r"\s*v\[(\d+):(\d+)\]", ops) if m.group(1).startswith("buffer") else None
        d0, d1 = (int(md.group(1)), int(md.group(2))) if md else (int(m.group(2)), int(m.group(3)))
        if d1 - d0 < 2:
            continue
        if count is not None:
            count[0] += 1
        waited = 0
        for nxt in lines[k + 1:k + 6]:
            t = nxt.split("//")[0]
            if re.match(r"^[0-9a-f]+ <", t) or waited >= 2:
                break
            if re.search(r"\b(s_branch|s_cbranch\w*|s_endpgm|s_setpc_b64)\b", t):
                break                                      # (what follows in the listing is not what follows in time)
            mn = re.search(r"s_nop\s+(\d+)", t)
            if mn:
                waited += int(mn.group(1)) + 1
                continue
            mv = VDEF.search(t)
            if mv and not mv.group(1).startswith("v_cmp"):
                defs = []
                if mv.group(2) is not None:
                    defs.append((int(mv.group(2)), int(mv.group(3))))
                elif mv.group(4) is not None:
                    defs.append((int(mv.group(4)), int(mv.group(4))))
                if "swap" in mv.group(1):
                    if mv.group(5) is not None:
                        defs.append((int(mv.group(5)), int(mv.group(6))))
                    elif mv.group(7) is not None:
                        defs.append((int(mv.group(7)), int(mv.group(7))))
                if any(a <= d1 and d0 <= b for a, b in defs):
                    out.append((func, text.strip(), t.strip()))
                    break
            if t.strip():
                waited += 1
    return out


MFMA_DST = re.compile(r"^\s*v_s?mfma\w*\s+([av])\[(\d+):(\d+)\]")
REGS = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+))\b")
EARLY_SLOTS = 11       # wait states between an MFMA and a non-MFMA instruction that touches its result (GCNHazardRecognizer: passes + 3: 7 for the 4-pass
                       # 16x16 shapes, 11 for the 8-pass 32x32 shapes); an instruction counts 1, s_nop N counts N + 1, an MFMA in between at least 4


def early_result_uses(lines):
    """(function, mfma, user) for non-MFMA instructions that touch an MFMA's destination registers within EARLY_SLOTS issue slots behind it (s_nop N
    counts N + 1).  hipcc keeps these distances for the builtins; an INLINE-ASM MFMA (csrc/rconv.hip, csrc/calib.hip) hides its latency from the
    compiler, which may then place a copy or a spill of the accumulator straight behind it -- wrong sums, silently (rounds 4 and 6)."""
    out, func = [], "?"
    recent = []                                            # (slots left, kind, lo, hi, text)
    for line in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func, recent = m.group(1), []
            continue
        text = line.split("//")[0]
        if ":" in text[:24]:
            text = text.split(":", 1)[1] if re.match(r"^\s*[0-9a-f]+:", text) else text
        t = text.strip()
        if not t:
            continue
        if re.search(r"\b(s_branch|s_cbranch\w*|s_endpgm|s_setpc_b64|s_barrier)\b", t):
            recent = []                                    # (what follows in the listing is not what follows in time; a barrier is long enough)
            continue
        mn = re.match(r"s_nop\s+(\d+)", t)
        md = MFMA_DST.match(t)
        step = int(mn.group(1)) + 1 if mn else 4 if md else 1
        if not md and not mn:
            for kind, a, b, one in REGS.findall(t.split(None, 1)[1] if len(t.split(None, 1)) > 1 else ""):
                lo, hi = (int(a), int(b)) if a else (int(one), int(one))
                for left, k2, l2, h2, txt in recent:
                    if k2 == kind and lo <= h2 and l2 <= hi:
                        out.append((func, txt, t))
                        break
        recent = [(left - step, k, l, h, x) for left, k, l, h, x in recent if left - step > 0]
        if md:
            recent.append((7 if "16x16" in t else EARLY_SLOTS, md.group(1), int(md.group(2)), int(md.group(3)), t))
    return out


def scan(obj):
    bad, total = [], 0
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}",
                        "--unbundle"], check=True, stderr=subprocess.DEVNULL)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    func = "?"
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func = m.group(1)
            continue
        m = PAT.search(line)
        if not m:
            continue
        total += 1
        dk, d0, d1, ck, c0, c1 = m.group(2), int(m.group(3)), int(m.group(4)), m.group(5), int(m.group(6)), int(m.group(7))
        if dk == ck and (d0, d1) != (c0, c1) and d0 <= c1 and c0 <= d1:
            bad.append((func, line.strip().split("//")[0].strip()))
    nst = [0]
    sth = store_hazards(dis.splitlines(), nst)
    return total, bad, sth, nst[0], early_result_uses(dis.splitlines())


# objects whose kernels are built on MFMAs and 16-byte stores: the scan must FIND both in them -- a disassembler whose text no longer matches the
# patterns above (or an empty build directory) must fail the check, not pass it vacuously
MFMA_OBJECTS = ("gemm.o", "gemm16.o", "ffn.o", "rowgemm.o", "attention.o", "attn64.o", "attn80.o", "tleg.o", "gnconv.o", "rconv.o")
NO_WIDE_STORES = ("attn64.o", "attn80.o")         # (its outputs leave as 8-byte stores)


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "mmgt_amd", "csrc", "build", "*.o")))
    rc = 0
    if not sys.argv[1:]:
        missing = [m for m in MFMA_OBJECTS if m not in {os.path.basename(o) for o in objs}]
        if missing:
            print(f"check_mfma_overlap: objects not built: {missing} (run make -C mmgt_amd/csrc first)")
            return 2
    for o in objs:
        total, bad, sth, nst, early = scan(o)
        print(f"{os.path.basename(o):24s} {total:6d} MFMAs, {len(bad)} with vDst partially overlapping SrcC; {nst} wide stores, {len(sth)} whose data is rewritten at once; "
              f"{len(early)} MFMA results touched within {EARLY_SLOTS} wait states")
        for f, mf, us in early[:20]:
            print(f"    {f[:60]}: {mf}   ->   {us}")
            rc = 1
        base = os.path.basename(o).replace("abl_", "")
        if base in MFMA_OBJECTS and (total == 0 or (nst == 0 and base not in NO_WIDE_STORES)):
            print(f"    PARSED NOTHING: {total} MFMAs / {nst} wide stores in an object that is built on them -- the disassembly no longer matches this tool's patterns")
            rc = 2
        for f, l in bad[:20]:
            print(f"    (harmless) {f[:80]}: {l}")
        for f, st, ov in sth[:20]:
            print(f"    {f[:60]}: {st}   <-   {ov}")
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
