#!/bin/bash
# Rebuild a kernel file's object from (possibly edited) device ISA: the host side is compiled from the source, the device side
# assembled from the .s (hipcc -S --cuda-device-only), bundled and embedded -- so that ISA-level edits can be timed / checked on
# the GPU without going through the register allocator again (finding 1).
#   tools/hazard/roundtrip.sh <source.hip> <device.s> <out.o>
set -e
LL=/opt/rocm/lib/llvm/bin
src=$1; asm=$2; out=$3
tmp=$(mktemp -d)
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$asm" -o $tmp/dev.o
$LL/ld.lld -shared $tmp/dev.o -o $tmp/dev.hsaco
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input=$tmp/dev.hsaco -output=$tmp/dev.hipfb
(cd "$(dirname "$src")" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value -Wno-pass-failed --cuda-host-only \
    -Xclang -fcuda-include-gpubinary -Xclang $tmp/dev.hipfb -c "$(basename "$src")" -o "$out")
rm -rf $tmp
echo "built $out"
