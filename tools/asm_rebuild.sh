#!/bin/bash
# asm_rebuild.sh <source dir> <file.hip> <device.s> <out.o>: host object of file.hip carrying the code object assembled from device.s
set -e
B=/opt/rocm/lib/llvm/bin
src=$1; hip=$2; s=$(readlink -f $3); out=$(readlink -f $4)
tmp=$(mktemp -d)
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $s -o $tmp/dev.o
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $tmp/dev.out $tmp/dev.o
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$tmp/dev.out -output=$tmp/dev.hipfb
(cd $src && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $tmp/dev.hipfb -c $hip -o $out)
rm -rf $tmp
