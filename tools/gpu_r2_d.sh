#!/bin/bash
set -u
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
O=gpurun_out/r2d
timeout 600 python tools/gemm_p8_bench.py > $O/gemm_p8_bench.log 2>&1
cat $O/gemm_p8_bench.log | tail -20
timeout 900 python -m pytest tests/test_gpu_path.py -m gpu -q -x --tb=short -p no:cacheprovider -k "eight_wave or tile_config or fused_band or golden_split or bf16_mode" > $O/tests.log 2>&1
tail -8 $O/tests.log
for s in 1 2; do
  for p8 in 0 -1; do
    VB_GEMM_P8=$p8 timeout 300 python bench.py --steps 2 --warmup 1 --streams $s --no-cpu-baseline > $O/bench_s${s}_p8_${p8}.json 2> $O/bench_s${s}_p8_${p8}.err
    python - <<PY
import json
d=json.loads([l for l in open('$O/bench_s${s}_p8_${p8}.json') if l.startswith('{')][-1])
print('streams', $s, 'p8', $p8, 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_check'] and d['parity_check']['ok'], [ (c['class'][:14], round(c['ms_per_pass'],1)) for c in d['roofline']['classes']])
PY
  done
done
