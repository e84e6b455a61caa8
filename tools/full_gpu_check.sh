#!/bin/bash
# What the driver runs at round end, in one GPU-box call, plus the suites on the deterministic library:  bash tools/full_gpu_check.sh
O=gpurun_out/full_check; mkdir -p $O
python -m pytest tests -q -m gpu > $O/full_gpu_tests.log 2>&1; echo "rc=$?" >> $O/full_gpu_tests.log; tail -4 $O/full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
RG_DETERMINISTIC=1 python -m pytest tests/test_parity_gpu.py tests/test_steps_gpu.py tests/test_kernels_gpu.py tests/test_config5_gpu.py tests/test_x3_gpu.py -q -p no:cacheprovider > $O/suites_on_the_deterministic_library.log 2>&1
echo "rc=$?" >> $O/suites_on_the_deterministic_library.log; tail -3 $O/suites_on_the_deterministic_library.log
