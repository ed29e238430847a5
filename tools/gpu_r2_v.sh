#!/bin/bash
# router tokens-per-wave A/B at full size
set -u
TAG=${1:-r2v}
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
for st in 1 2; do
for s in 4 2 1; do
  VB_ROUTER_TPW=$s timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --streams $st > $O/c2_s${st}_tpw$s.json 2> $O/c2_s${st}_tpw$s.err
  line c2_streams${st}_tpw$s $O/c2_s${st}_tpw$s.json
done
done
