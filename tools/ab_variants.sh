#!/bin/bash
# Build variants of ONE kernel source into separate libraries (recguru_amd/build/variants/<name>.so) for A/B timing in a
# single GPU-box call:  tools/ab_variants.sh attention name1=/path/to/variant1.hip name2=/path/to/variant2.hip
# then on the box:      RG_HIP_LIB=recguru_amd/build/variants/name1.so python tools/kb_attn.py
set -e
cd "$(dirname "$0")/.."
trap 'rm -f recguru_amd/csrc/_variant_*.hip' EXIT      # a failed hipcc must not leave a variant where build.py's *.hip glob finds it
src=$1; shift
python -m recguru_amd.build > /dev/null
mkdir -p recguru_amd/build/variants
others=$(ls recguru_amd/build/*.o | grep -v "/$src.o")
for kv in "$@"; do
  name=${kv%%=*}; file=${kv#*=}
  cp "$file" recguru_amd/csrc/_variant_$src.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value -Wno-pass-failed -c recguru_amd/csrc/_variant_$src.hip -o recguru_amd/build/variants/$name.o
  rm recguru_amd/csrc/_variant_$src.hip
  hipcc --offload-arch=gfx950 -shared -fPIC -o recguru_amd/build/variants/$name.so recguru_amd/build/variants/$name.o $others
  echo built recguru_amd/build/variants/$name.so
done
