#!/bin/bash
# round 6, call h: deeper weight ring of the two-per-CU minimal-filtering build - tests, layer microbench, end-to-end
set -u
mkdir -p gpurun_out/r06h
export TMPDIR=/tmp
O=gpurun_out/r06h
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "minimal_filtering" 2>&1 | tail -3
timeout 600 python tools/conv_mf_bench.py 8 > $O/mf_bench.txt 2>&1; tail -20 $O/mf_bench.txt
for i in 1 2; do
timeout 400 python bench.py --vocoder-precision fp32mf --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-isolated --detail $O/b.json 2> $O/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nsw5', d['value'], d['ms_per_step'], d['parity_check']['ok'])"
VB_MF_OCC=3 timeout 400 python bench.py --vocoder-precision fp32mf --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-isolated --detail $O/b3.json 2> $O/b3.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occ3/nsw3', d['value'], d['ms_per_step'], d['parity_check']['ok'])"
done
