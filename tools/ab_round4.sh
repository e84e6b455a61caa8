#!/bin/bash
# Round-4 A/B lines quoted in DESIGN 6 / 6a, one GPU-box call:  bash tools/ab_round4.sh  (writes gpurun_out/ab4/*.json and a summary)
mkdir -p gpurun_out/ab4
Q="--no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0"
run() { name=$1; shift; env "$@" python bench.py $BARGS $Q > gpurun_out/ab4/$name.json 2>> gpurun_out/ab4/err.log; }
BARGS="--steps 10 --warmup 3"
run bf16_default RG_NOOP=1
run bf16_no_tn_layer RG_NO_TN_LAYER=1
run bf16_tn_wgs192 RG_TN_LAYER_WGS=192
run bf16_tn_wgs384 RG_TN_LAYER_WGS=384
BARGS="--dtype bf16x3 --steps 4 --warmup 2"
run x3_default RG_NOOP=1
run x3_rt4 RG_X3_PA_RT=4
run x3_no_tn_layer RG_NO_TN_LAYER=1
run x3_ws768 RG_WS_SLOTS=768
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/ab4/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    ks = d["roofline"]["kernels_ms_per_step"]
    pick = {k: v for k, v in ks.items() if k.startswith(("gemm_tn", "post_attn", "gemm_ws_kernel<1,1>", "tn_"))}
    print("%-22s %9.1f seq/s %8.3f ms/step  %s" % (f.split("/")[-1][:-5], d["value"], d["ms_per_step"], pick))
PY
