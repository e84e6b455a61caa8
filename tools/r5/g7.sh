mkdir -p gpurun_out/r5b
for v in 0 1; do
RG_EMBED_OLD=$v python bench.py --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 --full_length_steps 0 --ae_steps 0 --steps 10 > gpurun_out/r5b/bench_embed_old$v.json 2> gpurun_out/r5b/bench_embed.err
done
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_dropout_gpu.py tests/test_fused256_gpu.py -q -x -k "embed or fused256 or post_attn256 or layer_fused" > gpurun_out/r5b/embed_tests2.log 2>&1
tail -2 gpurun_out/r5b/embed_tests2.log
python - <<'PY'
import json
for v in (0, 1):
    d = json.load(open("gpurun_out/r5b/bench_embed_old%d.json" % v))
    e = [o for o in d["roofline"]["other_kernels"] if o["kernel"].startswith("embed")][0]
    print("RG_EMBED_OLD=%d" % v, d["value"], d["ms_per_step"], "embed avg us", e["avg_launch_us"], "ms/step", d["roofline"]["kernels_ms_per_step"]["embed_pe_fwd_kernel"])
PY
