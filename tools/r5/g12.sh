mkdir -p gpurun_out/r5f
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batches_per_domain 1 --steps 1 --warmup 1 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 --tier_steps 0 --host_only_steps 0"
python bench.py --dtype bf16x3 $C5 > gpurun_out/r5f/c5_x3.json 2> gpurun_out/r5f/c5_x3.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5f/c5_x3.json"))
print(d["value"], d["ms_per_step"], d["config"]["last_step"]["recon_a"])
for k, v in list(d["roofline"]["kernels_ms_per_step"].items())[:24]: print("%-44s %8.1f" % (k, v))
PY
