mkdir -p gpurun_out/r5j
NB="--no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 --host_only_steps 0 --steps 4 --warmup 2 --dtype bf16x3"
for c in 1,1,1,1 1.2,1,1,1 1.4,1,1,1 0.85,1,1,1 1,1,1.15,1 1,1.15,1,1 1,1,1,1.2; do
  RG_TN_LAYER_COST=$c python bench.py $NB 2>/dev/null > gpurun_out/r5j/x3cost_$c.json
  python - "$c" <<'PY'
import json,sys
c=sys.argv[1]
b=json.load(open("gpurun_out/r5j/x3cost_%s.json"%c))
k=b["roofline"]["kernels_ms_per_step"]
print(c, b["ms_per_step"], "tn_layer", {n:v for n,v in k.items() if "tn_layer" in n})
PY
done
