#!/bin/bash
# attention (K-fragment prefetch, deferred rescale) + band kernel (fragment prefetch): tests, micro-benchmark, pipeline numbers
set -u
TAG=${1:-r2x}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py tests/test_gpu_production.py -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "attention or band or golden_split or sample_cfg or production or fullsize or bf16_mode" > $O/tests.log 2>&1
echo "tests exit: $?" >> $O/tests.log
tail -5 $O/tests.log
for thr in 8 0; do echo "VB_ATTN_DEFER=$thr"; VB_ATTN_DEFER=$thr timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_bench.txt; done
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    print(sys.argv[1], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'parity', d['parity_check'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for thr in 8 0; do
VB_ATTN_DEFER=$thr timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > $O/c2_thr$thr.json 2> $O/c2_thr$thr.err
line c2_defer$thr $O/c2_thr$thr.json
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-parity-check --streams 1 > $O/s1.log 2>&1
f=$(find $O/s1 -name "*kernel_stats.csv" | head -1); cp $f $O/s1_kernel_stats.csv
python $R/tools/prof_summary.py $O/s1_kernel_stats.csv 2 12
find $O -name "*kernel_trace.csv" -delete
