#!/bin/bash
set -u
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
O=gpurun_out/r2e
timeout 600 python tools/gemm_p8_bench.py > $O/gemm_p8_bench.log 2>&1
cat $O/gemm_p8_bench.log | tail -20
