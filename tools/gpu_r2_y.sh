#!/bin/bash
# GEMM ring variants inside the pipeline: 1 = BK64 x 2 stages (2 workgroups per CU), 3 = BK32 x 3 stages at 3 waves per SIMD (3 per CU)
set -u
export TMPDIR=/tmp
for v in 1 3 1 3 1 3; do
  VB_GEMM_VARIANT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v 2 streams', round(d['value'],1), d['parity_check']['ok'], d['device']['clocks_during_timed_region'].get('sclk_mhz_avg'))"
done
for v in 1 3; do
  VB_GEMM_VARIANT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v 1 stream', round(d['value'],1), d['parity_check']['ok'])"
done
VB_GEMM_VARIANT=3 python -m pytest tests/test_gpu_path.py tests/test_gpu_kernels.py -m gpu -q -x -k "gemm or golden_split or band or swiglu" -p no:cacheprovider 2>&1 | tail -2
