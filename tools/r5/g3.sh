mkdir -p gpurun_out/r5a
python tools/kb_embed_c5.py > gpurun_out/r5a/kb_embed_new2.txt 2>&1
RG_EMBED_OLD=1 python tools/kb_embed_c5.py > gpurun_out/r5a/kb_embed_old2.txt 2>&1
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_dropout_gpu.py -q -x -k "embed" > gpurun_out/r5a/embed_tests.log 2>&1
grep embed_pe gpurun_out/r5a/kb_embed_new2.txt gpurun_out/r5a/kb_embed_old2.txt; tail -3 gpurun_out/r5a/embed_tests.log
