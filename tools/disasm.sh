#!/bin/bash
# disassemble the gfx950 code of one built object:  tools/disasm.sh conv_igemm > /tmp/conv_igemm.s
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat vtoonify_amd/build/$1.o
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co
$L/llvm-objdump -d --no-show-raw-insn $T/co
rm -rf $T
