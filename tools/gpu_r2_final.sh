#!/bin/bash
# end-of-round evidence: whole GPU suite, smoke, the three bench workloads, b=1/b=2 lines, rocprofv3 kernel stats + PMC passes
set -u
TAG=${1:-r2z}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
O=gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "gpu tests exit: $?" >> $O/gpu_tests.log
tail -6 $O/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $O/smoke.log
timeout 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python bench.py --workload c3 --steps 1 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 400 python bench.py --workload c5 --steps 1 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --batch 2 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_b2.json 2> $O/bench_b2.err
timeout 300 python bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_s1.json 2> $O/bench_s1.err
for f in c2 c3 c5 b1 b2 s1; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/bench_$f.json') if l.startswith('{')][-1])
    print('$f', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_check'] and d['parity_check']['ok'], d['config']['sampler_loop'])
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
bash tools/gpu_prof_r2.sh $TAG/prof pmc 2>&1 | tail -40
