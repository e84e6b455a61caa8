#!/bin/bash
# The bit-level screens in one GPU-box call:  bash tools/check_bits.sh
#   1. tests/test_determinism_gpu.py (run-to-run at two workgroups per CU, chunk invariance, one vs two streams)
#   2. tools/fuzz_chunks.py (randomised chunk invariance with live-tile lists and ragged shapes)
#   3. per-launch checksums of whole steps at B = 1024 (d = 128) and at the config-5 shape
#   4. the fused block and the other kernels next to a second GPU process
#   0. (before anything runs) the static ISA screens of the built library: spills under a narrowed exec, packed-f32 src1 high-half selects
cd "$(dirname "$0")/.."
if ls recguru_amd/build/isa/*.s > /dev/null 2>&1; then python tools/isa_exec_screen.py recguru_amd/build/isa/*.s 2>&1 | tail -1
else echo "(static ISA screens: run by every build where the library is built; the kept ISA does not travel to this box)"; fi
python -m pytest tests/test_determinism_gpu.py -m gpu -q 2>&1 | tail -1
python tools/fuzz_chunks.py 40 2>&1 | tail -1
RG_BENCH_DROPOUT=0.5 RG_BENCH_MINLEN=199 RG_BENCH_B=1024 python tools/race_trace.py 3 bf16 2>&1 | grep "^run" | cut -c1-160
RG_BENCH_L=400 RG_BENCH_K=1024 RG_BENCH_V=2000000 RG_BENCH_D=256 RG_BENCH_DROPOUT=0.5 RG_BENCH_MINLEN=399 RG_BENCH_B=128 python tools/race_trace.py 3 bf16 2>&1 | grep "^run" | cut -c1-160
(python tools/repeat_steps.py 120 bf16 0 2 > /dev/null 2>&1 &)
sleep 8
python tools/race_post_attn.py 1500 2>&1 | tail -1
python tools/race_kernels.py 1500 2>&1 | tail -1
sleep 5
