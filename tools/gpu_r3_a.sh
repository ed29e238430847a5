#!/bin/bash
# round-3 opening measurement: baseline bench lines + kernel stats on this box, then the SQ counter pass
set -u
TAG=${1:-r3a}
mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
export TMPDIR=/tmp
timeout 500 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_s1.json 2> $O/bench_s1.err
timeout 300 python bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err
for f in c2 s1 b1; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/bench_$f.json') if l.startswith('{')][-1])
    print('$f', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_check'] and d['parity_check']['ok'])
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
bash tools/gpu_prof_r2.sh $TAG/prof 2>&1 | tail -60
bash tools/gpu_sq_pmc.sh $TAG 2>&1 | tail -40
