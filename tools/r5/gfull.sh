mkdir -p gpurun_out/r5e
python -m pytest tests -q -m gpu -x > gpurun_out/r5e/full_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5e/full_gpu_tests.log
tail -15 gpurun_out/r5e/full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5e/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r5e/smoke.log; tail -2 gpurun_out/r5e/smoke.log
