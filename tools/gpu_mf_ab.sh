#!/bin/bash
# Same-box A/B runs of the minimal-filtering fp32 convolution (round 6).  One gpurun call each:
#   bash tools/gpu_mf_ab.sh layers           tests + tools/conv_mf_bench.py (direct against minimal filtering per layer shape; VB_MF_OCC=3 once more)
#   bash tools/gpu_mf_ab.sh precisions       bench.py with fp32 (direct) and fp32mf as the primary precision, interleaved
#   bash tools/gpu_mf_ab.sh env NAME=V ...   bench.py --vocoder-precision fp32mf with and without the given environment, interleaved twice
#                                            (used for VB_FP32_PAIRS=32,64 / VB_MF_OCC=3 / VB_LRELU_IN_WINDOW=1: profiles/r06_mf_*.txt)
set -u
mkdir -p gpurun_out/mf_ab
export TMPDIR=/tmp
O=gpurun_out/mf_ab
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['parity_check']['ok'])"; }
bench() { timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-isolated --detail $O/d.json "$@" 2>> $O/err.log; }
case "${1:-layers}" in
  layers)
    timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "minimal_filtering" 2>&1 | tail -3
    timeout 600 python tools/conv_mf_bench.py 8 | tee $O/layers.txt
    VB_MF_OCC=3 timeout 600 python tools/conv_mf_bench.py 8 | tee $O/layers_occ3.txt ;;
  precisions)
    for i in 1 2; do for prec in fp32 fp32mf; do bench --vocoder-precision $prec | line $prec; done; done ;;
  env)
    shift
    for i in 1 2; do
      bench --vocoder-precision fp32mf | line default
      env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-isolated --detail $O/d.json --vocoder-precision fp32mf 2>> $O/err.log | line "$*"
    done ;;
esac
