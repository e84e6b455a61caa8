mkdir -p gpurun_out/r5b
python -m pytest tests/test_fused256_gpu.py -q > gpurun_out/r5b/fused256_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r5b/fused256_tests.log
python -m pytest tests/test_dropout_gpu.py tests/test_widths_gpu.py tests/test_determinism_gpu.py -q -x > gpurun_out/r5b/other_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r5b/other_tests.log
python bench.py --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 --steps 10 > gpurun_out/r5b/bench_short.json 2> gpurun_out/r5b/bench_short.err
tail -4 gpurun_out/r5b/fused256_tests.log; tail -4 gpurun_out/r5b/other_tests.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5b/bench_short.json"))
print(d["value"], d["ms_per_step"], list(d["roofline"]["kernels_ms_per_step"].items())[:12])
for o in d["roofline"]["other_kernels"]:
    if o["kernel"].startswith("embed"): print(o)
PY
