#!/bin/bash
# round 6, call b: the new under-load determinism tests + c3 / c5 bench lines with the current build
set -u
mkdir -p gpurun_out/r06b
export TMPDIR=/tmp
O=gpurun_out/r06b
timeout 1200 python -m pytest tests/test_gpu_production.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "beside or fused_glue" -rs 2>&1 | tail -15 > $O/load_tests.log
tail -6 $O/load_tests.log
timeout 400 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --detail $O/bench_c3.json > $O/bench_c3.line.json 2> $O/bench_c3.err
timeout 400 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --detail $O/bench_c5.json > $O/bench_c5.line.json 2> $O/bench_c5.err
for f in c3 c5; do wc -c $O/bench_$f.line.json; python - <<PY
import json
d=json.load(open('$O/bench_$f.json'))
print('$f', round(d['value'],1), round(d['ms_per_step'],2), d['parity_check'] and d['parity_check']['ok'], [ (g['group'][:20], round(g['ms_per_pass'],2), round(g['frac_of_mfma_peak'],3)) for g in d['roofline']['groups']])
PY
done
