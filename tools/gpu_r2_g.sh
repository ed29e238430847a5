#!/bin/bash
set -u
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
O=gpurun_out/r2g
run() { # name, env, args
  env $2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1])
print('$1', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'parity', d['parity_check'] and d['parity_check']['ok'])
PY
}
run s2 "A=1" "--streams 2"
run s4 "A=1" "--streams 4"
run s8 "A=1" "--streams 8"
run s2_nchunk6 "VB_GEMM_NCHUNK=6" "--streams 2"
run s2_nchunk9 "VB_GEMM_NCHUNK=9" "--streams 2"
run s1_nchunk6 "VB_GEMM_NCHUNK=6" "--streams 1"
run s1 "A=1" "--streams 1"
run b1 "A=1" "--streams 1 --batch 1"
run b2 "A=1" "--streams 1 --batch 2"
run b2s2 "A=1" "--streams 2 --batch 2"
run b4s4 "A=1" "--streams 4 --batch 4"
