#!/bin/bash
set -u
mkdir -p gpurun_out/r2n
export TMPDIR=/tmp
O=gpurun_out/r2n
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py tests/test_gpu_configs.py -m gpu -q -x --tb=short -p no:cacheprovider -k "staged or hifigan or fullsize or overlap_add or cli" > $O/tests.log 2>&1
tail -5 $O/tests.log
for e in 9 0; do
  VB_CONV_CFG=$e timeout 300 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > $O/bench_cfg$e.json 2> $O/bench_cfg$e.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_cfg$e.json') if l.startswith('{')][-1])
print('conv_cfg=$e', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'parity', d['parity_check'] and d['parity_check']['ok'], [ (c['class'][:14], round(c['ms_per_pass'],1), round(c['avg_launch_us'],1)) for c in d['roofline']['classes']])
PY
done
