mkdir -p gpurun_out/r5a
python -m pytest tests/test_determinism_gpu.py -q -x -k "row_wise" > gpurun_out/r5a/det_rowwise.log 2>&1
echo "rc=$?" >> gpurun_out/r5a/det_rowwise.log
python bench.py > gpurun_out/r5a/bench_default.json 2> gpurun_out/r5a/bench_default.err
echo "bench rc=$?"
tail -4 gpurun_out/r5a/det_rowwise.log; tail -3 gpurun_out/r5a/bench_default.err
