#!/bin/bash
# Round 6 A/B calls on the GPU box (each stage writes under gpurun_out/r06/<stage>/):
#   bash tools/ab_round6.sh nopk     : tools/peaks.py (tuned streams + co-execution fillers) and the packed-f32-free variant library
#                                      (tools/variants/v_nopk.so: `-target-feature -packed-fp32-ops` on the MFMA-carrying sources) against the shipped one
set -u
cd "$(dirname "$0")/.."
stage=${1:-nopk}
O=gpurun_out/r06/$stage; mkdir -p $O
QUICK="--no_cpu_baseline --tier_steps 0 --host_only_steps 0 --config5_steps 0 --ae_steps 0 --full_length_steps 0 --steps 20 --warmup 5"
ab_bench() {   # name lib [extra bench flags]
  local name=$1 libp=$2; shift 2
  if [ -n "$libp" ]; then RG_HIP_LIB=$libp python bench.py $QUICK "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  else python bench.py $QUICK "$@" > $O/bench_$name.json 2> $O/bench_$name.err; fi
  python - "$O/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks = j.get("roofline", {}).get("kernels", {})
    print(sys.argv[2], j["ms_per_step"], "ms", j["value"], "seq/s")
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1].get("ms_per_step", 0))[:14]:
        print("    %-44s %7.3f ms/step" % (k[:44], v.get("ms_per_step", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
case $stage in
nopk)
  python tools/peaks.py $O/peaks.txt > /dev/null 2> $O/peaks.err; grep -E "best|coexec|hbm_|valu_" $O/peaks.txt | cut -c1-400
  V=tools/variants/v_nopk.so
  ab_bench shipped_1 ""; ab_bench nopk_1 $V; ab_bench shipped_2 ""; ab_bench nopk_2 $V
  ab_bench shipped_x3 "" --dtype bf16x3 --steps 6 --warmup 2; ab_bench nopk_x3 $V --dtype bf16x3 --steps 6 --warmup 2
  for t in bf16 bf16x3; do
    python tools/kb_post_attn.py $t > $O/kb_post_attn_${t}_shipped.txt 2>&1; RG_HIP_LIB=$V python tools/kb_post_attn.py $t > $O/kb_post_attn_${t}_nopk.txt 2>&1
  done
  python tools/kb_attn_hm.py > $O/kb_attn_hm_shipped.txt 2>&1; RG_HIP_LIB=$V python tools/kb_attn_hm.py > $O/kb_attn_hm_nopk.txt 2>&1
  python tools/kb_ffn_bwd.py > $O/kb_ffn_bwd_shipped.txt 2>&1; RG_HIP_LIB=$V python tools/kb_ffn_bwd.py > $O/kb_ffn_bwd_nopk.txt 2>&1
  grep -H "us" $O/kb_*.txt | cut -c1-200
  # bits: the variant must pass the kernel / determinism / parity suites unchanged
  RG_HIP_LIB=$V timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py tests/test_parity_gpu.py tests/test_x3_gpu.py -q -x -p no:cacheprovider > $O/tests_nopk.log 2>&1
  tail -3 $O/tests_nopk.log
  ;;
embed)
  # the embedding gather taken apart one difference at a time (tools/embed_ladder.hip), the product kernel's two forms in the step
  tools/embed_ladder_probe > $O/embed_ladder.txt 2> $O/embed_ladder.err; cat $O/embed_ladder.txt | cut -c1-200
  ab_bench shipped ""; RG_EMBED_ROWS=1 ab_bench rows_form ""
  python tools/kb_embed_c5.py > $O/kb_embed_c5.txt 2>&1; RG_EMBED_ROWS=1 python tools/kb_embed_c5.py > $O/kb_embed_c5_rows.txt 2>&1; grep -h embed_pe $O/kb_embed_c5*.txt | cut -c1-220
  timeout 2400 python -m pytest tests/test_dp_hip_gpu.py -q -x -p no:cacheprovider -k "world_sizes or dp8 or eight" -s > $O/tests_dp8.log 2>&1; tail -12 $O/tests_dp8.log | cut -c1-300
  ;;
embed2)
  # the position-major embedding kernel (PE row in registers, nontemporal stores) against the old forms: stand-alone and in the step
  for cfg in "0 0" "2 0" "2 1"; do set -- $cfg
    RG_EMBED_FORM=$1 RG_EMBED_NT=$2 python tools/kb_embed_c5.py 2>&1 | grep embed_pe | sed "s/^/form=$1 nt=$2  /" | cut -c1-200 | tee -a $O/kb_embed_forms.txt
  done
  RG_EMBED_FORM=0 ab_bench form0 ""; RG_EMBED_FORM=2 RG_EMBED_NT=0 ab_bench form2 ""; RG_EMBED_FORM=2 RG_EMBED_NT=1 ab_bench form2_nt ""
  RG_EMBED_FORM=0 ab_bench form0_b ""; RG_EMBED_FORM=2 RG_EMBED_NT=1 ab_bench form2_nt_b ""
  timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_dropout_gpu.py tests/test_determinism_gpu.py -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
  ;;
shape)
  # the FFN structure of the fused block with both MFMA shapes (tools/ffn_shape_probe.hip); the embedding launcher's store policy
  tools/ffn_shape_probe_bin > $O/ffn_shape_probe.txt 2>&1; cat $O/ffn_shape_probe.txt | cut -c1-200
  python tools/kb_embed_c5.py 2>&1 | grep embed_pe | cut -c1-200 | tee $O/kb_embed_c5_auto_policy.txt
  ab_bench auto ""
  RG_EMBED_NT=0 ab_bench x3_nt0 "" --dtype bf16x3 --steps 6 --warmup 2; RG_EMBED_NT=1 ab_bench x3_nt1 "" --dtype bf16x3 --steps 6 --warmup 2
  QUICK="--no_cpu_baseline --tier_steps 0 --host_only_steps 0 --ae_steps 0 --full_length_steps 0 --steps 10 --warmup 3 --config5_steps 2"
  RG_EMBED_FORM=0 ab_bench c5_form0 ""; ab_bench c5_auto ""
  python - $O <<'PY'
import json, sys
for n in ("c5_form0", "c5_auto"):
    try:
        j = json.loads(open("%s/bench_%s.json" % (sys.argv[1], n)).read().strip().splitlines()[-1])
        c = j["config5"]
        print(n, "config5 ms/step", c.get("ms_per_step"), {k: v for k, v in c.get("kernels_ms_per_step", {}).items() if "embed_pe" in k})
    except Exception as e:
        print(n, "FAILED", e)
PY
  timeout 2400 python -m pytest tests/test_steps_gpu.py tests/test_det_gpu.py -q -x -p no:cacheprovider -k "entry_point or det" > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
  ;;
x3nt)
  # nontemporal stores for the plain-epilogue outputs of the weight-stationary GEMM in the f32-storage tiers (variant library v_wsnt.so)
  V=tools/variants/v_wsnt.so
  ab_bench x3_shipped_1 "" --dtype bf16x3 --steps 6 --warmup 2; ab_bench x3_wsnt_1 $V --dtype bf16x3 --steps 6 --warmup 2
  ab_bench x3_shipped_2 "" --dtype bf16x3 --steps 6 --warmup 2; ab_bench x3_wsnt_2 $V --dtype bf16x3 --steps 6 --warmup 2
  ;;
lastq)
  # the exact-f32 vector form of the x-input single-query attention (f32 / bf16x3 tiers): tests, then the tiers' step time with and without it
  timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "lastq" > $O/tests_lastq.log 2>&1; tail -5 $O/tests_lastq.log | cut -c1-300
  ab_bench x3_new "" --dtype bf16x3 --steps 6 --warmup 2; RG_NO_LASTQ_X=1 ab_bench x3_old "" --dtype bf16x3 --steps 6 --warmup 2
  ab_bench x3_new_b "" --dtype bf16x3 --steps 6 --warmup 2
  ab_bench f32_new "" --dtype f32 --steps 3 --warmup 1; RG_NO_LASTQ_X=1 ab_bench f32_old "" --dtype f32 --steps 3 --warmup 1
  timeout 2400 python -m pytest tests/test_steps_gpu.py tests/test_x3_gpu.py tests/test_parity_gpu.py tests/test_widths_gpu.py -q -x -p no:cacheprovider > $O/tests_tiers.log 2>&1; tail -5 $O/tests_tiers.log | cut -c1-300
  ;;
curves)
  # the bf16x3 loss-curve replay with and without the exact-f32 single-query kernels, three runs each (float-atomic order differs run to run)
  for i in 1 2 3; do
    python -m pytest tests/test_steps_gpu.py -q -p no:cacheprovider -k "loss_curves_replay and bf16x3" -s 2>&1 | grep "curves, bf16x3" | sed "s/^/new $i /" | cut -c1-420 | tee -a $O/curves.txt
    RG_NO_LASTQ_X=1 python -m pytest tests/test_steps_gpu.py -q -p no:cacheprovider -k "loss_curves_replay and bf16x3" -s 2>&1 | grep "curves, bf16x3" | sed "s/^/old $i /" | cut -c1-420 | tee -a $O/curves.txt
  done
  RG_DETERMINISTIC=1 python -m pytest tests/test_steps_gpu.py -q -p no:cacheprovider -k "loss_curves_replay and bf16x3" -s 2>&1 | grep "curves, bf16x3" | sed "s/^/new det /" | cut -c1-420 | tee -a $O/curves.txt
  RG_DETERMINISTIC=1 RG_NO_LASTQ_X=1 python -m pytest tests/test_steps_gpu.py -q -p no:cacheprovider -k "loss_curves_replay and bf16x3" -s 2>&1 | grep "curves, bf16x3" | sed "s/^/old det /" | cut -c1-420 | tee -a $O/curves.txt
  ;;
nopk_pa)
  # post_attn_fwd_kernel alone without packed-f32 instructions (per-function target attribute): bits + time
  timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py tests/test_x3_gpu.py tests/test_fused256_gpu.py tests/test_dropout_gpu.py -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-300
  for t in bf16 bf16x3; do python tools/kb_post_attn.py $t > $O/kb_post_attn_$t.txt 2>&1; done; grep -h us $O/kb_post_attn_*.txt
  ab_bench bf16_1 ""; ab_bench bf16_2 ""; ab_bench x3_1 "" --dtype bf16x3 --steps 6 --warmup 2
  ;;
esac
