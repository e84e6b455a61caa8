#!/bin/bash
# Round 6: everything profiles/r06/ holds, in one GPU-box call:  bash tools/profile_round6.sh
# New against tools/profile_round5.sh: the tuned peaks probe runs FIRST and its result is what every bench line of this call prices
# `roofline.achievable` against (same box); the embedding ladder, the FFN-shape probe, the single-query micro-benchmark of the f32-storage
# tiers; `bench.py --gpus 8` in its gloo-on-one-GPU form; a `train_gan.py --profile` sample.  Counter passes never share a run with a trace
# domain; the program itself stands behind `--`.
RD=r06
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$RD
rm -rf $O && mkdir -p $O
NB="--no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 --host_only_steps 0"
python3 $R/tools/peaks.py $O/peaks.txt > /dev/null 2>&1
mkdir -p $R/profiles/$RD && cp $O/peaks.txt $O/peaks.json $R/profiles/$RD/      # bench.py reads the newest profiles/rNN/peaks.json: THIS box's
(cd $R && tools/embed_ladder_probe > $O/embed_ladder.txt 2>&1; tools/ffn_shape_probe_bin > $O/ffn_shape_probe_32x32x16.txt 2>&1)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 4 --warmup 1 $NB > $O/bench_under_rocprof.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/kt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- python3 $R/bench.py --steps 1 --warmup 1 --no_roofline $NB > /dev/null 2> $O/pmcF.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- python3 $R/bench.py --steps 1 --warmup 1 --no_roofline $NB > /dev/null 2> $O/pmcW.err
(cd $R && python3 tools/pmc_traffic.py $O/pmcF $O/pmcW $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1)
rm -rf $O/pmcF $O/pmcW
mkdir -p $R/profiles/$RD && cp $O/pmc_traffic.json $R/profiles/$RD/pmc_traffic.json      # bench.py reads roofline.traffic from here
# ---- the bf16x3 tier: kernel stats, traffic counters, SQ counters
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x3kt -- python3 $R/bench.py --dtype bf16x3 --steps 3 --warmup 1 --no_roofline $NB > /dev/null 2> $O/x3kt.err
find $O/x3kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bf16x3.csv
rm -rf $O/x3kt
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF3 -- python3 $R/bench.py --dtype bf16x3 --steps 1 --warmup 1 --no_roofline $NB > /dev/null 2> $O/pmcF3.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW3 -- python3 $R/bench.py --dtype bf16x3 --steps 1 --warmup 1 --no_roofline $NB > /dev/null 2> $O/pmcW3.err
(cd $R && python3 tools/pmc_traffic.py $O/pmcF3 $O/pmcW3 $O/pmc_traffic_bf16x3.json > $O/pmc_summary_bf16x3.txt 2>&1)
rm -rf $O/pmcF3 $O/pmcW3
cd $R
rm -rf gpurun_out/pmc_pa gpurun_out/pmc_attn
bash tools/pmc_pa.sh train > $O/sq_post_attn.txt 2>&1
rm -rf gpurun_out/pmc_pa
bash tools/pmc_pa.sh train x3 > $O/sq_post_attn_bf16x3.txt 2>&1
rm -rf gpurun_out/pmc_pa
bash tools/pmc_attn.sh 0.5 > $O/sq_attention.txt 2>&1
rm -rf gpurun_out/pmc_attn
bash tools/pmc_attn.sh 0.5 x3 > $O/sq_attention_bf16x3.txt 2>&1
rm -rf gpurun_out/pmc_attn
# ---- the plain lines
timeout 900 python3 bench.py 2> $O/bench.err | tail -1 > $O/bench.json
timeout 300 python3 bench.py --dropout 0 --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 2>> $O/bench.err | tail -1 > $O/bench_dropout0.json
timeout 300 python3 bench.py --device_sampler --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 2>> $O/bench.err | tail -1 > $O/bench_device_sampler.json
timeout 300 python3 bench.py --mode ae --no_cpu_baseline --tier_steps 0 --config5_steps 0 --host_only_steps 0 2>> $O/bench.err | tail -1 > $O/bench_ae_step.json
timeout 600 python3 bench.py --dtype bf16x3 --steps 5 --warmup 2 --no_cpu_baseline --ae_steps 0 --full_length_steps 0 2>> $O/bench.err | tail -1 > $O/bench_bf16x3_tier.json
timeout 300 python3 tools/hostprof.py > $O/hostprof.txt 2>&1
timeout 300 python3 tools/hostprof2.py > $O/hostprof_torch_kernels.txt 2>&1
timeout 300 python3 tools/kb_embed_c5.py > $O/kb_embed_config5_table.txt 2>&1
RG_EMBED_FORM=0 timeout 300 python3 tools/kb_embed_c5.py > $O/kb_embed_config5_table_round5_kernel.txt 2>&1
timeout 300 python3 tools/kb_lastq.py bf16x3 > $O/kb_lastq_bf16x3.txt 2>&1
timeout 300 python3 tools/kb_lastq.py bf16 > $O/kb_lastq_bf16.txt 2>&1
for t in bf16 bf16x3; do timeout 300 python3 tools/kb_post_attn.py $t > $O/kb_post_attn_$t.txt 2>&1; done
timeout 300 python3 tools/kb_attn_hm.py > $O/kb_attn_hm.txt 2>&1
# `bench.py --gpus 8` as the driver's scaling run issues it, in the form a 1-GPU box can hold: eight rank processes on GPU 0 over gloo
RG_BENCH_SINGLE_DEVICE=1 RG_BENCH_BACKEND=gloo timeout 1200 python3 bench.py --gpus 8 --batch 64 --steps 2 --warmup 1 --no_cpu_baseline --batches_per_domain 1 --tier_steps 0 --ae_steps 0 --full_length_steps 0 --host_only_steps 0 2>> $O/bench.err | tail -1 > $O/bench_dp8_gloo_on_one_gpu.json
# the entry script's own per-kernel tables
rm -rf /tmp/rg_prof_res && timeout 900 python3 train_gan.py --cross True --synthetic 8192 --seq_len 200 --d_model 128 --n_head 4 --batch_size 4096 --batch_size_val 256 --vocab_size_a 100000 --vocab_size_b 100000 --n_negs 30 --phase1_steps 5 --steps_tune 10 --result_path /tmp/rg_prof_res --profile $O/train_gan_profile --profile_steps 3 > $O/train_gan_profile.log 2>&1
# ---- config-5 (2 M items, L = 400, d = 256, k = 1024) at B = 4096: with the fused d_model = 256 forward block and without
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batch 4096 --batches_per_domain 1 --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 --host_only_steps 0 --no_cpu_baseline"
mkdir -p $O/c5
timeout 900 python3 bench.py $C5 --steps 3 --warmup 1 2> $O/c5/bench.err | tail -1 > $O/c5/bench.json
RG_NO_PA256=1 timeout 900 python3 bench.py $C5 --steps 3 --warmup 1 2>> $O/c5/bench.err | tail -1 > $O/c5/bench_unfused_block.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5/kt -- python3 $R/bench.py $C5 --steps 2 --warmup 1 --no_roofline > $O/c5/bench_under_rocprof.json 2> $O/c5/kt.err)
find $O/c5/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5/kernel_stats.csv
rm -rf $O/c5/kt
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5/pmcF -- python3 $R/bench.py $C5 --steps 1 --warmup 1 --no_roofline > /dev/null 2> $O/c5/pmcF.err)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5/pmcW -- python3 $R/bench.py $C5 --steps 1 --warmup 1 --no_roofline > /dev/null 2> $O/c5/pmcW.err)
python3 tools/pmc_traffic.py $O/c5/pmcF $O/c5/pmcW $O/c5/pmc_traffic.json > $O/c5/pmc_summary.txt 2>&1
rm -rf $O/c5/pmcF $O/c5/pmcW
ls -la $O $O/c5
# ---- config-5 in the bf16x3 tier: the line and its kernel stats
timeout 900 python3 bench.py $C5 --dtype bf16x3 --steps 2 --warmup 1 2>> $O/c5/bench.err | tail -1 > $O/c5/bench_bf16x3.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5/kt3 -- python3 $R/bench.py $C5 --dtype bf16x3 --steps 1 --warmup 1 --no_roofline > /dev/null 2> $O/c5/kt3.err)
find $O/c5/kt3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5/kernel_stats_bf16x3.csv
rm -rf $O/c5/kt3
ls -la $O/c5
