mkdir -p gpurun_out/r5d
python -m pytest tests/test_fused256_gpu.py tests/test_config5_gpu.py -q -x > gpurun_out/r5d/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5d/tests.log
python -m pytest tests/test_kernels_gpu.py -q -x -k "item_loss or binned" > gpurun_out/r5d/tests_item.log 2>&1; echo "rc=$?" >> gpurun_out/r5d/tests_item.log
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batches_per_domain 1 --steps 2 --warmup 2 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 --tier_steps 0 --host_only_steps 0"
python bench.py $C5 > gpurun_out/r5d/c5_new.json 2> gpurun_out/r5d/c5_new.err
RG_ITEM_ONLINE_PLAIN=1 python bench.py $C5 > gpurun_out/r5d/c5_plain_online.json 2> gpurun_out/r5d/c5_plain.err
tail -3 gpurun_out/r5d/tests.log; tail -3 gpurun_out/r5d/tests_item.log
python - <<'PY'
import json
for f in ("c5_new", "c5_plain_online"):
    try:
        d = json.load(open("gpurun_out/r5d/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d["config"]["last_step"], list(d["roofline"]["kernels_ms_per_step"].items())[:6])
    except Exception as e:
        print(f, "failed", e)
PY
