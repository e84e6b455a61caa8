mkdir -p gpurun_out/r5j
NB="--no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 --host_only_steps 0 --steps 10 --warmup 3"
for c in 1.5,1,1,1 1.7,1,1,1 2.0,1,1,1 2.4,1,1,1 1.7,1,0.9,0.9 1.7,1,1.1,1.1 1.7,1.1,1,1 1.7,0.9,1,1 1.7,1,1,0.8 1.7,1,1,1.3; do
  RG_TN_LAYER_COST=$c python bench.py $NB 2>/dev/null > gpurun_out/r5j/cost_$c.json
  python - "$c" <<'PY'
import json,sys
c=sys.argv[1]
b=json.load(open("gpurun_out/r5j/cost_%s.json"%c))
k=b["roofline"]["kernels_ms_per_step"]
print(c, b["ms_per_step"], "tn_layer", k.get("gemm_tn_layer_kernel"))
PY
done
