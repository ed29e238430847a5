#!/bin/bash
set -u
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
O=gpurun_out/r2h
timeout 900 python -m pytest tests/test_gpu_path.py -m gpu -q -x --tb=short -p no:cacheprovider -k "graph_replay or determinism or eight_wave" > $O/tests.log 2>&1
tail -12 $O/tests.log
run() { # name, env, args
  env $2 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-isolated $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1])
print('$1', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'parity', d['parity_check'] and d['parity_check']['ok'], d['config']['sampler_loop'])
PY
}
run s2 "A=1" "--streams 2"
run s2_nograph "VB_NO_GRAPH=1" "--streams 2"
run s4 "A=1" "--streams 4"
run s8 "A=1" "--streams 8"
run s1 "A=1" "--streams 1"
run b1 "A=1" "--streams 1 --batch 1"
run b2 "A=1" "--streams 1 --batch 2"
run b2s2 "A=1" "--streams 2 --batch 2"
