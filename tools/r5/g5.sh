mkdir -p gpurun_out/r5b
python -m pytest tests/test_fused256_gpu.py -q -x > gpurun_out/r5b/fused256_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r5b/fused256_tests.log
python -m pytest tests/test_config5_gpu.py -q -x -s > gpurun_out/r5b/config5_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r5b/config5_tests.log
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batches_per_domain 1 --steps 2 --warmup 2 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 --tier_steps 0 --host_only_steps 0"
python bench.py $C5 > gpurun_out/r5b/c5_fused.json 2> gpurun_out/r5b/c5_fused.err
RG_NO_PA256=1 python bench.py $C5 > gpurun_out/r5b/c5_unfused.json 2> gpurun_out/r5b/c5_unfused.err
tail -5 gpurun_out/r5b/fused256_tests.log; tail -3 gpurun_out/r5b/config5_tests.log
python - <<'PY'
import json
for f in ("c5_fused", "c5_unfused"):
    try:
        d = json.load(open("gpurun_out/r5b/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], list(d["roofline"]["kernels_ms_per_step"].items())[:14])
    except Exception as e:
        print(f, "failed", e)
PY
