#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command -> gpurun_out/<tag>/kernel_stats.csv   (bash tools/rocprof_bench.sh <tag>)
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 4 --warmup 1 --no_cpu_baseline > $O/bench_under_rocprof.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/kt
head -40 $O/kernel_stats.csv | cut -c1-150
