mkdir -p gpurun_out/r5g
python -m pytest tests/test_widths_gpu.py -q -x -k "d256" > gpurun_out/r5g/tests_lists_x3.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/tests_lists_x3.log
python -m pytest tests/test_config5_gpu.py tests/test_fused256_gpu.py -q -x > gpurun_out/r5g/tests_c5b.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/tests_c5b.log
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batches_per_domain 1 --steps 1 --warmup 1 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 --tier_steps 0 --host_only_steps 0"
python bench.py --dtype bf16x3 $C5 > gpurun_out/r5g/c5_x3_lists.json 2> gpurun_out/r5g/c5_x3b.err
tail -4 gpurun_out/r5g/tests_lists_x3.log; tail -3 gpurun_out/r5g/tests_c5b.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g/c5_x3_lists.json"))
print(d["value"], d["ms_per_step"], d["config"]["last_step"]["recon_a"])
for k, v in list(d["roofline"]["kernels_ms_per_step"].items())[:12]: print("%-44s %8.1f" % (k, v))
PY
