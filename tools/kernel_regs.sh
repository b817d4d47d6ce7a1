#!/bin/bash
# Register / LDS / spill figures of every kernel in an object of mmgt_amd/csrc/build:  bash tools/kernel_regs.sh gemm16
L=/opt/rocm/lib/llvm/bin
o=mmgt_amd/csrc/build/$1.o
t=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin $o && \
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/fat.bin --output=$t/dev.co --unbundle 2>/dev/null && \
$L/llvm-readelf --notes $t/dev.co | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|vgpr_spill|\.sgpr_count|private_segment_fixed" | paste - - - - - - | sed 's/  */ /g' | cut -c1-260
rm -rf $t
