#!/bin/bash
# XCD-aware block mapping of the weight-stationary GEMM, on / off (RG_WS_NO_XCD=1), config-5 and the bf16x3 tier: bash tools/ab_ws_xcd.sh
mkdir -p gpurun_out/ab_xcd
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batch 4096 --batches_per_domain 1 --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 --no_cpu_baseline"
Q="--no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0"
for v in 0 1; do
  RG_WS_NO_XCD=$v python bench.py $C5 --steps 2 --warmup 2 > gpurun_out/ab_xcd/c5_noxcd$v.json 2>> gpurun_out/ab_xcd/err.log
  RG_WS_NO_XCD=$v python bench.py --dtype bf16x3 --steps 4 --warmup 2 $Q > gpurun_out/ab_xcd/x3_noxcd$v.json 2>> gpurun_out/ab_xcd/err.log
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_xcd/*.json")):
    d = json.load(open(f))
    ks = d["roofline"]["kernels_ms_per_step"]
    print("%-16s %9.1f seq/s %9.3f ms/step  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], {k: v for k, v in ks.items() if k.startswith("gemm_ws")}))
PY
