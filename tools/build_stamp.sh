#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (-DRG_STAMP) -> recguru_amd/build/librecguru_stamp.so (tools/stamp_*.py)
cd "$(dirname "$0")/../recguru_amd/csrc" || exit 1
mkdir -p ../build/stamp
for f in *.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value -Wno-pass-failed -DRG_STAMP -c $f -o ../build/stamp/${f%.hip}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/librecguru_stamp.so ../build/stamp/*.o && echo built
