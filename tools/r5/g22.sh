mkdir -p gpurun_out/r5i
timeout 1500 python -m pytest tests/test_dp_hip_gpu.py -q -s -rs -k "bench or rccl" > gpurun_out/r5i/bench_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5i/bench_tests.log
tail -15 gpurun_out/r5i/bench_tests.log
RG_DP_FORCE=1 timeout 600 python bench.py --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 --ae_steps 0 --full_length_steps 0 > gpurun_out/r5i/bench_rccl_group_of_one.json 2> gpurun_out/r5i/bench_rccl1.err; echo "rc=$?"
wc -l gpurun_out/r5i/bench_rccl_group_of_one.json
python - <<'PY'
import json
b=json.load(open("gpurun_out/r5i/bench_rccl_group_of_one.json"))
print(b["value"], b["ms_per_step"], b.get("exchange"))
PY
