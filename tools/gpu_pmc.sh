#!/bin/bash
# PMC passes on the GEMM micro-benchmark (the full bench crashes rocprofv3 --pmc under python on this image)
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc/$c -o g -- python $R/tools/gemm_bench.py > $R/gpurun_out/pmc/$c.log 2>&1
  echo "$c exit $?"
done
cd $R
find gpurun_out/pmc -name "*.csv" | head; for f in $(find gpurun_out/pmc -name "*counter_collection*.csv"); do echo $f; head -3 $f; wc -l $f; done
tail -3 gpurun_out/pmc/FETCH_SIZE.log
