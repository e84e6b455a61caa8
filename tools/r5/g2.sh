mkdir -p gpurun_out/r5a
python -m pytest tests/test_steps_gpu.py -q -s -k "bench_shape_loss_curve and (mixed or bf16x3)" > gpurun_out/r5a/mixed_curve.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5a/mixed_curve.log
python tools/kb_embed_c5.py > gpurun_out/r5a/kb_embed_new.txt 2>&1
RG_EMBED_OLD=1 python tools/kb_embed_c5.py > gpurun_out/r5a/kb_embed_old.txt 2>&1
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -q -x -k "embed" > gpurun_out/r5a/embed_tests.log 2>&1
python bench.py --dtype mixed --steps 5 --warmup 2 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 > gpurun_out/r5a/bench_mixed_v1.json 2> gpurun_out/r5a/bench_mixed_v1.err
grep -E "^\[|passed|failed" gpurun_out/r5a/mixed_curve.log; cat gpurun_out/r5a/kb_embed_new.txt gpurun_out/r5a/kb_embed_old.txt; tail -3 gpurun_out/r5a/embed_tests.log
