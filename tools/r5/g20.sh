mkdir -p gpurun_out/r5i
timeout 900 python -m pytest tests/test_dp_hip_gpu.py::test_rccl_group_of_one_rank tests/test_det_gpu.py::test_rccl_group_of_one_rank_is_bit_identical_to_no_process_group -q -s -rs > gpurun_out/r5i/rccl1.log 2>&1; echo "rc=$?" >> gpurun_out/r5i/rccl1.log
tail -25 gpurun_out/r5i/rccl1.log
RG_DP_FORCE=1 timeout 600 python bench.py --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 --ae_steps 0 --full_length_steps 0 2> gpurun_out/r5i/bench_rccl1.err | tail -1 > gpurun_out/r5i/bench_rccl_group_of_one.json
python - <<'PY'
import json
b=json.load(open("gpurun_out/r5i/bench_rccl_group_of_one.json"))
print(b["value"], b["ms_per_step"], b.get("exchange"))
PY
tail -5 gpurun_out/r5i/bench_rccl1.err
