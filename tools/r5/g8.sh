mkdir -p gpurun_out/r5c
python -m pytest tests/test_fused256_gpu.py -q -x > gpurun_out/r5c/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5c/tests.log
python tools/kb_post_attn.py > gpurun_out/r5c/kb_pa_4wave.txt 2>&1
RG_PA8=1 python tools/kb_post_attn.py > gpurun_out/r5c/kb_pa_8wave.txt 2>&1
for v in 0 1; do
RG_PA8=$v python bench.py --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 --full_length_steps 0 --ae_steps 0 --steps 10 > gpurun_out/r5c/bench_pa8_$v.json 2>> gpurun_out/r5c/bench.err
done
tail -3 gpurun_out/r5c/tests.log; grep "encoder, inference" gpurun_out/r5c/kb_pa_4wave.txt gpurun_out/r5c/kb_pa_8wave.txt
python - <<'PY'
import json
for v in (0, 1):
    d = json.load(open("gpurun_out/r5c/bench_pa8_%d.json" % v))
    print("RG_PA8=%d" % v, d["value"], d["ms_per_step"], {k: x for k, x in list(d["roofline"]["kernels_ms_per_step"].items())[:5]})
PY
