for s in 768 512 384; do
RG_WS_SLOTS=$s python bench.py --dtype bf16x3 --steps 4 --warmup 2 --no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 > gpurun_out/x3_ws_$s.json 2>gpurun_out/x3_ws.err
python - <<PY
import json
d=json.load(open("gpurun_out/x3_ws_$s.json"))
ks=d["roofline"]["kernels_ms_per_step"]
print("slots $s", d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "gemm_ws" in k})
PY
done
