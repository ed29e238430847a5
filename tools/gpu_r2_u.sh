#!/bin/bash
# c2 A/B: small-tile threshold, number of streams
set -u
TAG=${1:-r2u}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
O=gpurun_out/$TAG
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    print(sys.argv[1], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'parity', d['parity_check'] and d['parity_check']['ok'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for rep in 1 2; do
for s in 300 160; do
  VB_GEMM_SMALL_TILES=$s timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > $O/c2_t$s.$rep.json 2> $O/c2_t$s.$rep.err
  line c2_tiles$s.$rep $O/c2_t$s.$rep.json
done
done
for n in 3 4; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --streams $n > $O/c2_s$n.json 2> $O/c2_s$n.err
line c2_streams$n $O/c2_s$n.json
done
