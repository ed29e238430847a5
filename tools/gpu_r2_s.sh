#!/bin/bash
# small-batch GEMM tiles, one-token-per-wave router, 8-expert router loads / K=96 DMA kernel: tests, micro-benchmark, bench lines
set -u
TAG=${1:-r2s}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
O=gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "router or bucket or swiglu or gemm or graph_replay or c3 or c1 or band or golden_split or sample_cfg" > $O/tests.log 2>&1
echo "tests exit: $?" >> $O/tests.log
tail -5 $O/tests.log
timeout 200 python tools/gemm_tilecfg.py small 2>&1 | tee $O/tilecfg_small.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    print(sys.argv[1], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'parity', d['parity_check'] and d['parity_check']['ok'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for s in 0 11 21; do
  VB_GEMM_SMALL=$s timeout 300 python bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-isolated > $O/b1_small$s.json 2> $O/b1_small$s.err
  line b1_small$s $O/b1_small$s.json
done
for s in 0 11; do
  VB_GEMM_SMALL=$s timeout 300 python bench.py --batch 2 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-isolated > $O/b2_small$s.json 2> $O/b2_small$s.err
  line b2_small$s $O/b2_small$s.json
done
timeout 400 python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated > $O/c3.json 2> $O/c3.err
line c3 $O/c3.json
timeout 400 python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err
line c2 $O/c2.json
