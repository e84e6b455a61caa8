mkdir -p gpurun_out/r5h
python -m pytest tests -q -m gpu > gpurun_out/r5h/full_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5h/full_gpu_tests.log
tail -8 gpurun_out/r5h/full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5h/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r5h/smoke.log; tail -2 gpurun_out/r5h/smoke.log
