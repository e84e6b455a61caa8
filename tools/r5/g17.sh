mkdir -p gpurun_out/r5h
RG_ATTN_BWD_X3_RESTAGE=1 python -m pytest tests/test_x3_gpu.py -q -x -k "attention" > gpurun_out/r5h/tests_attn_restage.log 2>&1; echo "rc=$?" >> gpurun_out/r5h/tests_attn_restage.log
for v in 0 1; do
if [ $v = 1 ]; then export RG_ATTN_BWD_X3_RESTAGE=1; else unset RG_ATTN_BWD_X3_RESTAGE; fi
python bench.py --dtype bf16x3 --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 --full_length_steps 0 --ae_steps 0 --steps 5 --warmup 2 > gpurun_out/r5h/bench_x3_restage$v.json 2>> gpurun_out/r5h/bench.err
done
tail -3 gpurun_out/r5h/tests_attn_restage.log
python - <<'PY'
import json
for v in (0, 1):
    d = json.load(open("gpurun_out/r5h/bench_x3_restage%d.json" % v))
    print("restage=%d" % v, d["value"], d["ms_per_step"], {k: x for k, x in list(d["roofline"]["kernels_ms_per_step"].items())[:8]})
PY
