#!/bin/bash
set -u
mkdir -p gpurun_out/r2j
export TMPDIR=/tmp
O=gpurun_out/r2j
run() { # name, env, args
  env $2 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1])
print('$1', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'parity', d['parity_check'] and d['parity_check']['ok'])
PY
}
for v in 1 4 2 3 0; do
  run b1_v$v "VB_GEMM_VARIANT=$v" "--streams 1 --batch 1"
done
for v in 1 4 2; do
  run b2_v$v "VB_GEMM_VARIANT=$v" "--streams 1 --batch 2"
done
for v in 4 2; do
  run b8_v$v "VB_GEMM_VARIANT=$v" "--streams 2"
done
run b1_tile33 "VB_GEMM_TILE=33" "--streams 1 --batch 1"
