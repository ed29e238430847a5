#!/bin/bash
set -u
TAG=${1:-r3e2}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
O=gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "gpu tests exit: $?" >> $O/gpu_tests.log
tail -8 $O/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-pmc > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python bench.py --workload c5 --steps 1 --warmup 1 --no-pmc > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --batch 2 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_b2.json 2> $O/bench_b2.err
for f in c2 c5 b1 b2; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/bench_$f.json') if l.startswith('{')][-1])
    print('$f', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_check'] and d['parity_check']['ok'], d['device']['clocks_during_timed_region'].get('sclk_mhz_avg'))
    for r in d['roofline']['classes']: print('   ', r['class'][:44], round(r['ms_per_pass'],2), 'ms', round(r['avg_launch_us'],1),'us', round(r['frac_of_mfma_peak'],4), r['launches_per_pass'])
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
