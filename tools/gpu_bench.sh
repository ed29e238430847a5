#!/bin/bash
# staged bench probe: small -> full, each stage under its own timeout; logs in gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cat /sys/fs/cgroup/cpu.max 2>/dev/null | tee gpurun_out/cpu.log; nproc | tee -a gpurun_out/cpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 200 python bench.py --batch 1 --flow-steps 2 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -12 | tee gpurun_out/bench_small.log
timeout 300 python bench.py --batch 8 --flow-steps 50 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -12 | tee gpurun_out/bench_b8_nocpu.log
timeout 420 python bench.py --steps 1 --warmup 1 2>&1 | tail -12 | tee gpurun_out/bench.log
