#!/bin/bash
# counter passes over tools/pa_only.py (one group per run, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_pa/g$i -- python3 $R/tools/pa_only.py ${1:-train} 3 $2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_pa/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "post_attn" not in k: continue
        acc["post_attn"][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    for c, v in sorted(d.items()):
        print("%-32s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
