#!/bin/bash
# rocprofv3 kernel-trace stats (+ optional PMC passes) of the default bench command
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $R/gpurun_out/prof/trace.log 2>&1
tail -3 $R/gpurun_out/prof/trace.log
if [ "${1:-}" = "pmc" ]; then
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof/fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $R/gpurun_out/prof/fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof/write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $R/gpurun_out/prof/write.log 2>&1
fi
cd $R
find gpurun_out/prof -name "*stats*" | head
# keep only small summaries (the raw traces are large)
find gpurun_out/prof -name "*kernel_trace*.csv" -size +20M -delete
ls -la gpurun_out/prof/*/* | head -30
