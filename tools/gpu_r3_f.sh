#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_multirank.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for a in "--batch 1 --streams 1 --steps 4 --warmup 2" "--batch 2 --streams 1 --steps 4 --warmup 2" "--steps 3 --warmup 1"; do
  timeout 300 python bench.py --no-cpu-baseline --no-pmc $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', round(d['value'],1), round(d['ms_per_step'],2), d['parity_check']['ok'], [ (r['class'][:10], round(r['ms_per_pass'],2)) for r in d['roofline']['classes']])"
done
