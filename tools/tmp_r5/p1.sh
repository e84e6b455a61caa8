mkdir -p gpurun_out/r5p
timeout 600 python -m pytest tests/test_fused256_gpu.py -q -k "pipelined" > gpurun_out/r5p/t.log 2>&1; tail -3 gpurun_out/r5p/t.log
NB="--no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 --host_only_steps 0 --steps 10 --warmup 3"
for v in 1 0 1 0; do RG_PA_PIPE=$v python bench.py $NB 2>/dev/null > gpurun_out/r5p/bench_pipe$v.json; python - $v <<'PY'
import json,sys
b=json.load(open("gpurun_out/r5p/bench_pipe%s.json"%sys.argv[1]))
print("PIPE",sys.argv[1], b["ms_per_step"], b["roofline"]["kernels_ms_per_step"]["post_attn_fwd_kernel<bf16>"], b["roofline"]["avg_launch_us"])
PY
done
