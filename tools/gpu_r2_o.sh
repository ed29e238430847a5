#!/bin/bash
set -u
mkdir -p gpurun_out/r2o
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q -x --tb=short -p no:cacheprovider -k "router or golden_split or bf16_mode or determinism or graph_replay" > $O/tests.log 2>&1
tail -4 $O/tests.log
cd /tmp
VB_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -o b -- python $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-isolated > $O/s1.log 2>&1
f=$(find $O/s1 -name "*kernel_stats.csv" | head -1)
python $R/tools/prof_summary.py $f 2 16
find $O -name "*kernel_trace.csv" -delete
cd $R
