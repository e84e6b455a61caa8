mkdir -p gpurun_out/r5g
timeout 1500 python -m pytest tests/test_det_gpu.py -x -q -s > gpurun_out/r5g/det.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/det.log
tail -40 gpurun_out/r5g/det.log
