#!/bin/bash
# same-box A/B of environment knobs on the default two-stream bench value (REPS runs each, default 3, interleaved)
#   bash tools/gpu_ab_value.sh "<env A or ->" "<env B>" ...
set -u
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-3}); do
  for envs in "$@"; do
    if [ "$envs" = "-" ]; then e=""; else e="$envs"; fi
    v=$(env $e timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-isolated --steps 3 --warmup 1 ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['parity_check']['ok'])")
    echo "rep $rep [$envs] $v"
  done
done
