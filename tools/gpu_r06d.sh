#!/bin/bash
# round 6, call d: minimal filtering end to end - tests, then bench with fp32 (direct) and fp32mf as the primary precision, same box
set -u
mkdir -p gpurun_out/r06d
export TMPDIR=/tmp
O=gpurun_out/r06d
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "minimal_filtering" 2>&1 | tail -5 > $O/mf_tests.log
tail -3 $O/mf_tests.log
for prec in fp32 fp32mf fp32 fp32mf; do
  timeout 400 python bench.py --vocoder-precision $prec --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --detail $O/bench_$prec.json > $O/bench_$prec.line.json 2> $O/bench_$prec.err
  python - <<PY
import json
d=json.load(open('$O/bench_$prec.json'))
print('$prec', round(d['value'],1), round(d['ms_per_step'],2), d['parity_check'] and d['parity_check']['ok'], d['parity_check'] and d['parity_check'].get('mel_l1_sampled'), [(g['group'][:12], round(g['ms_per_pass'],2), round(g['frac_of_mfma_peak'],3)) for g in d['roofline']['groups']])
PY
done
VB_MF_OCC=2 timeout 400 python bench.py --vocoder-precision fp32mf --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-isolated --detail $O/bench_occ2.json 2> $O/bench_occ2.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occ2', d['value'], d['ms_per_step'])"
