#!/bin/bash
# refresh the secondary bench lines (c3, c5, batch 1, batch 2) under gpurun_out/$1
set -u
O=gpurun_out/${1:-lines}; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python bench.py --workload c3 --steps 1 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 400 python bench.py --workload c5 --steps 1 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --batch 2 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_b2.json 2> $O/bench_b2.err
for f in c3 c5 b1 b2; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/bench_$f.json') if l.startswith('{')][-1])
    print('$f', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_check'] and d['parity_check']['ok'], 'traffic', d['roofline']['traffic'])
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
