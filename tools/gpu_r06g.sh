#!/bin/bash
# round 6, call g: per-kernel profile of the one-stream fp32mf pass
set -u
TAG=r06g
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-pmc --no-parity-check --streams 1 --vocoder-precision fp32mf --detail $O/d.json > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/stream1_fp32mf_kernel_stats.csv
python $R/tools/prof_summary.py $O/stream1_fp32mf_kernel_stats.csv 3 40
find $O/prof -name "*kernel_trace.csv" -delete
