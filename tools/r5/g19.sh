mkdir -p gpurun_out/r5g
RG_DETERMINISTIC=1 timeout 1700 python -m pytest tests/test_parity_gpu.py tests/test_steps_gpu.py tests/test_kernels_gpu.py tests/test_config5_gpu.py tests/test_x3_gpu.py -q -p no:cacheprovider > gpurun_out/r5g/det_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/det_suite.log
tail -30 gpurun_out/r5g/det_suite.log
python - <<'PY' >> gpurun_out/r5g/det_suite.log 2>&1
import os
os.environ["RG_DETERMINISTIC"]="1"
PY
