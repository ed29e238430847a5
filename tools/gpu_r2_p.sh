#!/bin/bash
set -u
mkdir -p gpurun_out/r2q
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2q
timeout 600 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/multirank.log 2>&1
tail -3 $O/multirank.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_c2.json') if l.startswith('{')][-1])
print('c2', round(d['value'],1), round(d['ms_per_step'],2), d['parity_check']['ok'], d['config']['sampler_loop'], d['device'], d['roofline']['traffic'], d['roofline']['frac'])
PY
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o b -- python $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --streams 1 > $O/c3.log 2>&1
f=$(find $O/c3 -name "*kernel_stats.csv" | head -1); cp $f $O/c3_kernel_stats.csv
python $R/tools/prof_summary.py $O/c3_kernel_stats.csv 3 16
find $O -name "*kernel_trace.csv" -delete
cd $R
