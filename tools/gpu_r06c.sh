#!/bin/bash
# round 6, call c: minimal-filtering fp32 convolution - parity tests and the layer micro-benchmark (both occupancy builds)
set -u
mkdir -p gpurun_out/r06c
export TMPDIR=/tmp
O=gpurun_out/r06c
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "minimal_filtering" 2>&1 | tail -25 > $O/mf_tests.log
tail -12 $O/mf_tests.log
timeout 600 python tools/conv_mf_bench.py 8 > $O/mf_bench_occ3.txt 2>&1; tail -22 $O/mf_bench_occ3.txt
VB_MF_OCC=2 timeout 600 python tools/conv_mf_bench.py 8 > $O/mf_bench_occ2.txt 2>&1; tail -3 $O/mf_bench_occ2.txt
