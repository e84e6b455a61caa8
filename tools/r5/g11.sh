mkdir -p gpurun_out/r5f
python -m pytest tests/test_widths_gpu.py tests/test_config5_gpu.py tests/test_fused256_gpu.py -q -x > gpurun_out/r5f/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5f/tests.log
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batches_per_domain 1 --steps 2 --warmup 2 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 --tier_steps 0 --host_only_steps 0"
python bench.py $C5 > gpurun_out/r5f/c5_lists.json 2> gpurun_out/r5f/c5.err
RG_NO_LISTS_256=1 python bench.py $C5 > gpurun_out/r5f/c5_nolists.json 2>> gpurun_out/r5f/c5.err
tail -3 gpurun_out/r5f/tests.log
python - <<'PY'
import json
for f in ("c5_lists", "c5_nolists"):
    try:
        d = json.load(open("gpurun_out/r5f/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d["config"]["last_step"]["recon_a"], list(d["roofline"]["kernels_ms_per_step"].items())[:16])
    except Exception as e:
        print(f, "failed", e)
PY
