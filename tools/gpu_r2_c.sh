#!/bin/bash
# round 2, call 3: full GPU suite after the host-side changes + the new API / at-size tests + bench lines for c2 / c3 / c5
set -u
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
O=gpurun_out/r2c
timeout 2400 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "gpu tests exit: $?" >> $O/gpu_tests.log
tail -15 $O/gpu_tests.log
timeout 400 python bench.py --steps 2 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 3000 $O/bench_c2.json; tail -3 $O/bench_c2.err
timeout 400 python bench.py --workload c3 --steps 1 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
tail -c 1500 $O/bench_c3.json; tail -3 $O/bench_c3.err
timeout 400 python bench.py --workload c5 --steps 1 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
tail -c 1500 $O/bench_c5.json; tail -3 $O/bench_c5.err
