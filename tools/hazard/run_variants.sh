#!/bin/bash
# tools/race_post_attn.py (M = 65400 = 327 sequences of 200: two workgroups per CU) on every variant library tools/hazard/v_*.so
# (tools/hazard/isa_variants.py) and on the shipped library:  bash tools/hazard/run_variants.sh [launches per form]
for f in tools/hazard/v_*.so ""; do
  echo "== ${f:-shipped library}"
  RG_HIP_LIB=$f RG_RACE_M=65400 timeout 300 python tools/race_post_attn.py ${1:-10} 2>&1 | grep "launches differ\|total differing"
done
