#!/bin/bash
# Round 5 timing-only ablations of the attention kernels (bounds on the levers, measured):
#   no_ds    : the one-pass backward without its in-sweep dQ path (dS scratch write into the pad columns, ds_read_tr16_b64, 16-deep MFMAs)
#              -- an upper bound on what ANY re-layout of the dS transpose (swizzled scratch tile, fewer bank conflicts) can return
#   no_dmask : the forward without the per-head hash of the dropout words -- the largest single item of its per-head prologue
# Builds recguru_amd/build/variants/{no_ds,no_dmask}.so here; on the GPU box: bash tools/ab_round5.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" != "run" ]; then
  (echo "#define RG_ABL_NO_DS 1"; cat recguru_amd/csrc/attention.hip) > /tmp/attn_no_ds.hip
  (echo "#define RG_ABL_NO_DMASK 1"; cat recguru_amd/csrc/attention.hip) > /tmp/attn_no_dmask.hip
  bash tools/ab_variants.sh attention no_ds=/tmp/attn_no_ds.hip no_dmask=/tmp/attn_no_dmask.hip
  mkdir -p tools/variants && cp recguru_amd/build/variants/no_ds.so tools/variants/v_no_ds.so && cp recguru_amd/build/variants/no_dmask.so tools/variants/v_no_dmask.so   # (build/variants is not sent to the GPU box)
  exit 0
fi
O=gpurun_out/ab_r5
mkdir -p $O
python tools/kb_attn.py 0:0.5 1:0.5 > $O/attn_shipped.txt 2>&1
RG_ALLOW_UNSCREENED=1 RG_HIP_LIB=tools/variants/v_no_ds.so python tools/kb_attn.py 0:0.5 1:0.5 > $O/attn_no_ds.txt 2>&1
RG_ALLOW_UNSCREENED=1 RG_HIP_LIB=tools/variants/v_no_dmask.so python tools/kb_attn.py 0:0.5 1:0.5 > $O/attn_no_dmask.txt 2>&1
python tools/kb_attn_hm.py > $O/attn_hm_shipped.txt 2>&1
RG_ALLOW_UNSCREENED=1 RG_HIP_LIB=tools/variants/v_no_dmask.so python tools/kb_attn_hm.py > $O/attn_hm_no_dmask.txt 2>&1
grep -H "causal\|us" $O/*.txt | head -40
