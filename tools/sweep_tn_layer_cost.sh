#!/bin/bash
# Sweep of the per-slot cost factors of rg_gemm_tn_layer's workgroup split (hip.LAYER_COST, RG_TN_LAYER_COST), one GPU-box call:
#   bash tools/sweep_tn_layer_cost.sh            -> profiles/r05/ab/tn_layer_cost_sweep.txt was written from its output
O=gpurun_out/tn_layer_cost; mkdir -p $O
NB="--no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 --host_only_steps 0"
one() {   # file tag, extra bench args, kernel name
  python - "$1" "$3" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], b["ms_per_step"], {n: v for n, v in b["roofline"]["kernels_ms_per_step"].items() if sys.argv[2] in n})
PY
}
for c in 1,1,1,1 1.15,1,1,1 1.3,1,1,1 1.45,1,1,1 1.5,1,1,1 1.7,1,1,1 2.0,1,1,1 1.45,1.1,1,1 1.45,1,1.1,1 1.45,1,1,1.2; do
  RG_TN_LAYER_COST=$c python bench.py $NB --steps 10 --warmup 3 2>/dev/null > $O/bf16_$c.json; one $O/bf16_$c.json "" tn_layer
done
for c in 1,1,1,1 1.2,1,1,1 0.85,1,1,1 1,1.15,1,1 1,1,1.15,1 1,1,1,1.2; do
  RG_TN_LAYER_COST=$c python bench.py $NB --steps 4 --warmup 2 --dtype bf16x3 2>/dev/null > $O/x3_$c.json; one $O/x3_$c.json "" tn_layer
done
