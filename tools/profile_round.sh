#!/bin/bash
# Everything profiles/<round>/ holds, in one GPU-box call:  bash tools/profile_round.sh [round-dir-name]
# (kernel trace + stats of the bench command, the two --pmc traffic passes, the plain bench lines incl. full-length users,
# host-side profile, micro-benchmarks, SQ counter passes on the dominant kernels).  Counter passes never share a run with a
# trace domain.
RD=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$RD
rm -rf $O && mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 4 --warmup 1 --no_cpu_baseline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 > $O/bench_under_rocprof.json 2> $O/kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 > /dev/null 2> $O/pmcF.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 > /dev/null 2> $O/pmcW.err
cd $R
python3 tools/pmc_traffic.py $O/pmcF $O/pmcW $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1
mkdir -p profiles/$RD && cp $O/pmc_traffic.json profiles/$RD/pmc_traffic.json      # bench.py reads roofline.traffic from here
timeout 600 python3 bench.py 2> $O/bench.err | tail -1 > $O/bench.json
timeout 300 python3 bench.py --dropout 0 --no_cpu_baseline --tier_steps 0 --config5_steps 0 2>> $O/bench.err | tail -1 > $O/bench_dropout0.json
timeout 300 python3 bench.py --device_sampler --no_cpu_baseline --tier_steps 0 --config5_steps 0 2>> $O/bench.err | tail -1 > $O/bench_device_sampler.json
timeout 300 python3 bench.py --min_len 199 --no_cpu_baseline --tier_steps 0 --config5_steps 0 2>> $O/bench.err | tail -1 > $O/bench_full_length_users.json
timeout 300 python3 bench.py --batch 32 --steps 20 --no_cpu_baseline --no_roofline --tier_steps 0 --config5_steps 0 2>> $O/bench.err | tail -1 > $O/bench_tiny_batch_host_only.json
timeout 300 python3 tools/hostprof.py > $O/hostprof.txt 2>&1
timeout 300 python3 tools/hostprof2.py > $O/hostprof_torch_kernels.txt 2>&1
timeout 300 python3 tools/kb_disc.py > $O/kb_disc.txt 2>&1
timeout 300 python3 tools/kb_embed_c5.py > $O/kb_embed_config5_table.txt 2>&1
timeout 300 python3 tools/kb_attn.py > $O/kb_attention.txt 2>&1
timeout 300 python3 tools/kb_tn.py > $O/kb_weight_gradient_gemms.txt 2>&1
RG_TN_REGSTAGE=1 timeout 300 python3 tools/kb_tn.py > $O/kb_weight_gradient_gemms_register_staged.txt 2>&1
timeout 300 python3 tools/kb_ffn_bwd.py > $O/kb_ffn_backward.txt 2>&1
timeout 300 python3 tools/kb_item_loss.py > $O/kb_item_loss.txt 2>&1
timeout 300 python3 tools/kb_lastq.py > $O/kb_lastq.txt 2>&1
timeout 300 python3 tools/kb_embed_scatter.py > $O/kb_embed_scatter.txt 2>&1
bash tools/pmc_pa.sh train > $O/sq_post_attn.txt 2>&1
bash tools/pmc_attn.sh 0.5 > $O/sq_attention.txt 2>&1
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
# config-5 (2 M items, L = 400, d = 256, k = 1024) at B = 4096: bench line + kernel trace
C5="--items 2000000 --seq_len 400 --d_model 256 --n_head 8 --n_negs 1024 --batch 4096 --batches_per_domain 1 --ae_steps 0 --full_length_steps 0 --tier_steps 0 --config5_steps 0 --no_cpu_baseline"
mkdir -p $O/c5
timeout 900 python3 bench.py $C5 --steps 3 --warmup 1 2> $O/c5/bench.err | tail -1 > $O/c5/bench.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5/kt -- python3 $R/bench.py $C5 --steps 2 --warmup 1 --no_roofline > $O/c5/bench_under_rocprof.json 2> $O/c5/kt.err)
find $O/c5/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5/kernel_stats.csv
rm -rf $O/c5/kt
# config-5 traffic counters (separate FETCH_SIZE / WRITE_SIZE passes, never with a trace domain)
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5/pmcF -- python3 $R/bench.py $C5 --steps 1 --warmup 1 --no_roofline > /dev/null 2> $O/c5/pmcF.err)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5/pmcW -- python3 $R/bench.py $C5 --steps 1 --warmup 1 --no_roofline > /dev/null 2> $O/c5/pmcW.err)
python3 tools/pmc_traffic.py $O/c5/pmcF $O/c5/pmcW $O/c5/pmc_traffic.json > $O/c5/pmc_summary.txt 2>&1
rm -rf $O/c5/pmcF $O/c5/pmcW
timeout 600 python3 bench.py --mode ae --no_cpu_baseline --tier_steps 0 --config5_steps 0 2>> $O/bench.err | tail -1 > $O/bench_ae_step.json
timeout 600 python3 bench.py --residual split --no_cpu_baseline --ae_steps 0 --full_length_steps 0 2>> $O/bench.err | tail -1 > $O/bench_split_residual.json
timeout 900 python3 bench.py --dtype f32 --steps 3 --warmup 1 --no_cpu_baseline --ae_steps 0 --full_length_steps 0 2>> $O/bench.err | tail -1 > $O/bench_f32_tier.json
timeout 900 python3 bench.py --dtype bf16x3 --steps 5 --warmup 2 --no_cpu_baseline --ae_steps 0 --full_length_steps 0 2>> $O/bench.err | tail -1 > $O/bench_bf16x3_tier.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x3kt -- python3 $R/bench.py --dtype bf16x3 --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --ae_steps 0 --full_length_steps 0 > /dev/null 2> $O/x3kt.err)
find $O/x3kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bf16x3.csv
rm -rf $O/x3kt
timeout 300 python3 tools/kb_attn_hm.py > $O/kb_attention_head_major.txt 2>&1
timeout 300 python3 tools/kb_post_attn.py > $O/kb_post_attn.txt 2>&1
rm -rf $O/kt $O/pmcF $O/pmcW $R/gpurun_out/pmc_pa $R/gpurun_out/pmc_attn
ls -la $O
