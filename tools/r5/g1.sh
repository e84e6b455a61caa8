mkdir -p gpurun_out/r5a
python tools/peaks.py gpurun_out/r5a/peaks.txt > gpurun_out/r5a/peaks.log 2>&1
python -m pytest tests/test_steps_gpu.py -x -q -s -k "mixed or (bench_shape and bf16x3)" > gpurun_out/r5a/mixed_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5a/mixed_tests.log
python bench.py --dtype mixed --steps 5 --warmup 2 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 > gpurun_out/r5a/bench_mixed_v1.json 2> gpurun_out/r5a/bench_mixed_v1.err
python bench.py --dtype bf16x3 --steps 5 --warmup 2 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 > gpurun_out/r5a/bench_x3.json 2> gpurun_out/r5a/bench_x3.err
tail -3 gpurun_out/r5a/peaks.log; tail -30 gpurun_out/r5a/mixed_tests.log
