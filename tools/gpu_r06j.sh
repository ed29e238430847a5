#!/bin/bash
# round 6, call j: occupancy choice of the minimal-filtering kernel per launch (VB_MF_OCC = 2 / 3 / 0 = by rounds), default two-stream bench and one stream
set -u
mkdir -p gpurun_out/r06j
export TMPDIR=/tmp
O=gpurun_out/r06j
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --vocoder-precision fp32mf --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-isolated --detail $O/$tag.json $EXTRA 2> $O/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', '$EXTRA', d['value'], d['ms_per_step'], d['parity_check']['ok'])"; }
for rep in 1 2; do
EXTRA="" run occ2 VB_MF_OCC=2
EXTRA="" run occ3 VB_MF_OCC=3
EXTRA="" run auto VB_MF_OCC=0
done
EXTRA="--streams 1" run occ2s1 VB_MF_OCC=2
EXTRA="--streams 1" run occ3s1 VB_MF_OCC=3
EXTRA="--streams 1" run autos1 VB_MF_OCC=0
