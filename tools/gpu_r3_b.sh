#!/bin/bash
# round-3 check after the housekeeping / bench-verification changes: whole GPU suite + the default bench line (incl. live PMC)
set -u
TAG=${1:-r3b}
mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "gpu tests exit: $?" >> $O/gpu_tests.log
tail -15 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
tail -12 $O/bench_c2.err
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_c2.json') if l.startswith('{')][-1])
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'])
print(json.dumps(d['parity_check'])[:1500])
PY
