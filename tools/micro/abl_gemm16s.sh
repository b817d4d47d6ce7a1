#!/bin/bash
# Timing ablations of gemm16s_kernel: `make -C mmgt_amd/csrc abl` builds one library per G16S_ABL value (1 no LDS-DMA, 2 no vmcnt waits,
# 4 no fragment reads, 8 no MFMAs, 16 no epilogue rows; sums combine); this runs tools/ab_cfg.py's g16s shapes under each, gemm16_kernel beside.
# Results of the ablated builds are wrong by construction.   usage: bash tools/abl_gemm16s.sh > profiles/r4/abl_gemm16s.txt
set -e
cd "$(dirname "$0")/.."
for v in "" _abl2 _abl1 _abl4 _abl8 _abl16 _abl7 _abl15 _abl31; do
  echo "## libmmgt_hip$v.so  (c1 = gemm16_kernel, c2 = gemm16s_kernel)"
  MMGT_LIB=mmgt_amd/libmmgt_hip$v.so AB=g16_ver:1,2 SET=g16s ROUNDS=3 timeout -k 10 100 python tools/ab_cfg.py 2>&1 | grep -v amdgpu.ids | sed 's/ d=[0-9.e+-]*//g'
done
