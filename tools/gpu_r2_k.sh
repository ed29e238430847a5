#!/bin/bash
set -u
mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
O=gpurun_out/r2k
run() { # name, env, args
  env $2 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1])
print('$1', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'parity', d['parity_check'] and d['parity_check']['ok'])
PY
}
run b1_fused "A=1" "--streams 1 --batch 1"
run b1_unfused "VB_BAND_UNFUSED=1" "--streams 1 --batch 1"
run b2_fused "A=1" "--streams 1 --batch 2"
run b2_unfused "VB_BAND_UNFUSED=1" "--streams 1 --batch 2"
run b4_fused "A=1" "--streams 1 --batch 4"
run b4_unfused "VB_BAND_UNFUSED=1" "--streams 1 --batch 4"
run b8s2_fused "A=1" "--streams 2"
run b8s2_unfused "VB_BAND_UNFUSED=1" "--streams 2"
run b8s1_unfused "VB_BAND_UNFUSED=1" "--streams 1"
