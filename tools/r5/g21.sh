mkdir -p gpurun_out/r5i
RG_DP_FORCE=1 timeout 600 python -X faulthandler bench.py --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 --ae_steps 0 --full_length_steps 0 > gpurun_out/r5i/bench_rccl1.out 2> gpurun_out/r5i/bench_rccl1.err; echo "rc=$?" >> gpurun_out/r5i/bench_rccl1.err
tail -30 gpurun_out/r5i/bench_rccl1.err; tail -c 600 gpurun_out/r5i/bench_rccl1.out
