#!/bin/bash
# round 6, call i: LeakyReLU between a ResBlock's convolutions in the first one's epilogue - vocoder tests, then the A/B (both orders)
set -u
mkdir -p gpurun_out/r06i
export TMPDIR=/tmp
O=gpurun_out/r06i
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_api.py -m gpu -q -x -p no:cacheprovider -k "hifigan or vocode or fullsize or bigvgan or fp32_vocoder" 2>&1 | tail -4
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-isolated --detail $O/$tag.json 2> $O/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['parity_check']['ok'], d['fp32_direct']['value'], d['split']['value'])"; }
run epi X=1
run window VB_LRELU_IN_WINDOW=1
run epi2 X=1
run window2 VB_LRELU_IN_WINDOW=1
