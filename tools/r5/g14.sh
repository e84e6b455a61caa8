mkdir -p gpurun_out/r5g
python -m pytest tests/test_x3_gpu.py -q -x -k "attention" > gpurun_out/r5g/tests_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/tests_attn.log
python -m pytest tests/test_config5_gpu.py -q -x -s > gpurun_out/r5g/tests_c5.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/tests_c5.log
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batches_per_domain 1 --steps 1 --warmup 1 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 --tier_steps 0 --host_only_steps 0"
python bench.py --dtype bf16x3 $C5 > gpurun_out/r5g/c5_x3_restage.json 2> gpurun_out/r5g/c5_x3.err
tail -4 gpurun_out/r5g/tests_attn.log; grep "^\[config-5\|passed\|failed" gpurun_out/r5g/tests_c5.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g/c5_x3_restage.json"))
print(d["value"], d["ms_per_step"], d["config"]["last_step"])
for k, v in list(d["roofline"]["kernels_ms_per_step"].items())[:10]: print("%-44s %8.1f" % (k, v))
PY
