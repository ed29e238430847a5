#!/bin/bash
# same-box A/B of environment knobs: rocprofv3 kernel stats of the one-stream bench per setting, selected kernels printed side by side
#   bash tools/gpu_ab.sh <tag> "<grep pattern of kernels>" "ENV1=a ENV2=b" "ENV1=c" ...     (use - for the default environment)
set -u
TAG=$1; PAT=$2; shift 2
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
i=0
for envs in "$@"; do
  i=$((i+1))
  cd /tmp
  if [ "$envs" = "-" ]; then e=""; else e="$envs"; fi
  env $e timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab$i -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated --no-pmc --streams 1 ${BENCH_ARGS:-} > $O/ab$i.log 2>&1
  cd $R
  f=$(find $O/ab$i -name "*kernel_stats.csv" | head -1)
  echo "== [$envs]  $(grep -o '"value": [0-9.]*' $O/ab$i.log | head -1)  $(grep -o '"ok": [a-z]*' $O/ab$i.log | head -1)"
  python tools/prof_summary.py "$f" 4 40 | grep -E "total kernel|$PAT"
  cp "$f" $O/ab${i}_kernel_stats.csv
  find $O/ab$i -name "*kernel_trace.csv" -delete
done
