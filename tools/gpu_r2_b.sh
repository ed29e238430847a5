#!/bin/bash
set -u
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout 600 python tools/batch_invariance.py > gpurun_out/r2b/batch_invariance.log 2>&1
tail -20 gpurun_out/r2b/batch_invariance.log
