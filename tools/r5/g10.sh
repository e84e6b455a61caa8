mkdir -p gpurun_out/r5d
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batches_per_domain 1 --steps 2 --warmup 2 --no_cpu_baseline --full_length_steps 0 --ae_steps 0 --tier_steps 0 --host_only_steps 0"
python bench.py $C5 > gpurun_out/r5d/c5_ppw64k.json 2> gpurun_out/r5d/c5.err
RG_PPW_WIDE=262144 python bench.py $C5 > gpurun_out/r5d/c5_ppw256k.json 2>> gpurun_out/r5d/c5.err
RG_PPW_WIDE=1048576 python bench.py $C5 > gpurun_out/r5d/c5_ppw1m.json 2>> gpurun_out/r5d/c5.err
python -m pytest tests/test_kernels_gpu.py tests/test_config5_gpu.py -q -x -k "binned or item_loss or config5" > gpurun_out/r5d/tests2.log 2>&1; echo "rc=$?" >> gpurun_out/r5d/tests2.log
tail -3 gpurun_out/r5d/tests2.log
python - <<'PY'
import json
for f in ("c5_ppw64k", "c5_ppw256k", "c5_ppw1m"):
    try:
        d = json.load(open("gpurun_out/r5d/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d["config"]["last_step"]["recon_a"], list(d["roofline"]["kernels_ms_per_step"].items())[:4])
    except Exception as e:
        print(f, "failed", e)
PY
