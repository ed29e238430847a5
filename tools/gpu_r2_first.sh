#!/bin/bash
# round 2, first call: parity of the benchmarked precision, the self-spawning bench, a baseline bench line
set -u
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
O=gpurun_out/r2a
timeout 1500 python -m pytest tests/test_gpu_production.py -m gpu -q -rA --tb=short -s -p no:cacheprovider > $O/prod.log 2>&1
echo "prod exit: $?" >> $O/prod.log
tail -30 $O/prod.log
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q -rA --tb=short -s -p no:cacheprovider > $O/multirank.log 2>&1
echo "multirank exit: $?" >> $O/multirank.log
tail -30 $O/multirank.log
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_b8.json 2> $O/bench_b8.err
tail -2 $O/bench_b8.json
timeout 300 python bench.py --steps 3 --warmup 1 --batch 1 --streams 1 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err
tail -2 $O/bench_b1.json
