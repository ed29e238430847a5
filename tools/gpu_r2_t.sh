#!/bin/bash
# batch-1 kernel profile of the current build
set -u
TAG=${1:-r2t}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated --streams 1 --batch 1 > $O/b1.log 2>&1
f=$(find $O/b1 -name "*kernel_stats.csv" | head -1)
cp "$f" $O/b1_kernel_stats.csv
tail -1 $O/b1.log | cut -c1-200
python $R/tools/prof_summary.py $O/b1_kernel_stats.csv 3 40
find $O -name "*kernel_trace.csv" -delete
