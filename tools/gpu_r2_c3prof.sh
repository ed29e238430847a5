#!/bin/bash
# kernel stats of the c3 workload (8 experts, 32 clips) with the final build
set -u
O=$PWD/gpurun_out/c3prof; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o b -- python $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-parity-check --streams 1 > $O/c3.log 2>&1
f=$(find $O/c3 -name "*kernel_stats.csv" | head -1); cp $f $O/c3_kernel_stats.csv
python $R/tools/prof_summary.py $O/c3_kernel_stats.csv 3 14
find $O -name "*kernel_trace.csv" -delete
