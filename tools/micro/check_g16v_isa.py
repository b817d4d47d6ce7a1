#!/usr/bin/env python3
"""Build-time check of gemm16v.hip's register allocation (run by __graft_entry__.build() and the CPU test suite): the kernel's inline-asm MFMAs hide
their latency from hipcc, so it is only correct while every accumulator tile keeps ONE register quad for the whole kernel -- round 4 saw hipcc
time-share a quad between several tiles through scratch when all 256 accumulation registers were asked for.  Compiles the file to ISA and asserts,
per instantiation: no scratch instruction, every v_mfma destination == its own source C, 64 distinct destination quads each written by the same
number of MFMAs.   usage: python tools/check_g16v_isa.py   (exit code 0 = ok)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, "mmgt_amd", "csrc", "gemm16v.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "g16v.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src),
               "-S", "--cuda-device-only", "-Wno-unused-result", "-o", out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    kernels = re.split(r"^(_ZN\S*gemm16v_kernel\S*):", text, flags=re.M)[1:]
    bad = []
    n = 0
    for name, body in zip(kernels[0::2], kernels[1::2]):
        body = body.split("s_endpgm")[0]
        n += 1
        if re.search(r"\bscratch_", body):
            bad.append((name, "scratch instructions (spills)"))
        dst = collections.Counter()
        for m in re.finditer(r"v_mfma_f32_16x16x32_bf16 (\S+), \S+, \S+, (\S+)", body):
            d, c = m.group(1).rstrip(","), m.group(2)
            if d != c:
                bad.append((name, f"MFMA destination {d} != source C {c}"))
            dst[d] += 1
        if len(dst) != 64 or len(set(dst.values())) != 1:
            bad.append((name, f"{len(dst)} accumulator quads, MFMAs per quad {sorted(set(dst.values()))} (expected 64 quads, one count)"))
    if n != 14:
        bad.append(("*", f"{n} instantiations found (expected 14)"))
    for b in bad:
        print("gemm16v ISA check FAILED:", *b)
    if not bad:
        print(f"gemm16v ISA check: {n} instantiations, 64 accumulator quads each in place, no scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
