#!/bin/bash
# fused caption-gate scores + router: bit-equality tests, pipeline A/B, kernel time
set -u
TAG=${1:-r2sr}; O=$PWD/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "score_router or router or golden_split or graph_replay" 2>&1 | tail -15
for u in 0 1 0 1; do
  if [ $u = 1 ]; then export VB_SCORE_UNFUSED=1; else unset VB_SCORE_UNFUSED; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused=$u 2 streams', round(d['value'],1), d['parity_check']['ok'])"
done
for u in 0 1; do
  if [ $u = 1 ]; then export VB_SCORE_UNFUSED=1; else unset VB_SCORE_UNFUSED; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused=$u 1 stream', round(d['value'],1), d['parity_check']['ok'])"
done
unset VB_SCORE_UNFUSED
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-parity-check --streams 1 > $O/s1.log 2>&1
f=$(find $O/s1 -name "*kernel_stats.csv" | head -1); cp $f $O/s1_kernel_stats.csv
grep -E "score_router|router_kernel|glds_kernel<1" $O/s1_kernel_stats.csv | cut -d, -f1-4
find $O -name "*kernel_trace.csv" -delete
