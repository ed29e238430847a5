#!/bin/bash
# round 6, call e: 64-channel minimal-filtering tile - tests, microbench rows, and the end-to-end A/B of fused 64-channel pairs against unfused mf launches
set -u
mkdir -p gpurun_out/r06e
export TMPDIR=/tmp
O=gpurun_out/r06e
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "minimal_filtering" 2>&1 | tail -5 > $O/mf_tests.log
tail -3 $O/mf_tests.log
timeout 600 python tools/conv_mf_bench.py 8 2>&1 | grep "Ci=  64\|sum" | tee $O/mf_bench64.txt
run() { # tag env...
  tag=$1; shift
  env "$@" timeout 400 python bench.py --vocoder-precision fp32mf --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --detail $O/bench_$tag.json > $O/bench_$tag.line.json 2> $O/bench_$tag.err
  python - <<PY
import json
d=json.load(open('$O/bench_$tag.json'))
print('$tag', round(d['value'],1), round(d['ms_per_step'],2), d['parity_check'] and d['parity_check']['ok'], [(g['group'][:12], round(g['ms_per_pass'],2), round(g['frac_of_mfma_peak'],3)) for g in d['roofline']['groups']])
PY
}
run pairs3264 VB_MF_OCC=2 VB_FP32_PAIRS=32,64
run pairs32 VB_MF_OCC=2 VB_FP32_PAIRS=32
run pairs3264b VB_MF_OCC=2 VB_FP32_PAIRS=32,64
run pairs32b VB_MF_OCC=2 VB_FP32_PAIRS=32
