#!/bin/bash
set -u
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
O=gpurun_out/r2f
for s in 1 2; do
  for p8 in 0 -1; do
    VB_GEMM_P8=$p8 timeout 300 python bench.py --steps 3 --warmup 1 --streams $s --no-cpu-baseline > $O/bench_s${s}_p8_${p8}.json 2> $O/bench_s${s}_p8_${p8}.err
    python - <<PY
import json
d=json.loads([l for l in open('$O/bench_s${s}_p8_${p8}.json') if l.startswith('{')][-1])
print('streams', $s, 'p8', $p8, 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_check'] and d['parity_check']['ok'], [ (c['class'][:14], round(c['ms_per_pass'],1)) for c in d['roofline']['classes']])
PY
  done
done
