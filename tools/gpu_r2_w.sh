#!/bin/bash
# QKV epilogue ablations inside the pipeline (one stream, kernel stats)
set -u
TAG=${1:-r2w}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
cd /tmp
for a in 0 6 7 8; do
VB_GEMM_ABLATE=$a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/a$a -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-parity-check --streams 1 > $O/a$a.log 2>&1
f=$(find $O/a$a -name "*kernel_stats.csv" | head -1)
echo "ablate $a: $(grep 'gemm_bf16_glds_kernel<2' $f | cut -d, -f1-4 | cut -c1-120)"
echo "          $(grep 'attn_kernel' $f | cut -d, -f1-4 | cut -c1-120)"
find $O -name "*kernel_trace.csv" -delete
done
