#!/bin/bash
# fused band-expert kernel for 96-channel bands (8 experts): bit-equality, c3 bench A/B, kernel time
set -u
O=$PWD/gpurun_out/b96; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_configs.py tests/test_gpu_production.py -m gpu -q -x --tb=short -p no:cacheprovider -k "band or c3" 2>&1 | tail -8
for u in 0 1 0 1; do
  if [ $u = 1 ]; then export VB_BAND_UNFUSED=1; else unset VB_BAND_UNFUSED; fi
  python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 band_unfused=$u', round(d['value'],1), round(d['ms_per_step'],1), d['parity_check']['ok'])"
done
unset VB_BAND_UNFUSED
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o b -- python $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-parity-check --streams 1 > $O/c3.log 2>&1
f=$(find $O/c3 -name "*kernel_stats.csv" | head -1); cp $f $O/c3_kernel_stats.csv
python $R/tools/prof_summary.py $O/c3_kernel_stats.csv 3 10
find $O -name "*kernel_trace.csv" -delete
