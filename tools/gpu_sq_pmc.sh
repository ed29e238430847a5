#!/bin/bash
# SQ counter pass of the bench (one stream, whole batch of 8): MFMA-busy / wait breakdown per kernel.
#   bash tools/gpu_sq_pmc.sh [tag] [extra bench args]
# Writes gpurun_out/<tag>/sq_summary.json (copy to profiles/ to keep).  Counters only with --kernel-trace (gpurun rule).
set -u
TAG=${1:-sq}
shift || true
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d $R/gpurun_out/$TAG/sq -o b -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-pmc --no-parity-check "$@" > $R/gpurun_out/$TAG/sq.log 2>&1
echo "sq pass exit $?"
cd $R
python tools/sq_summary.py gpurun_out/$TAG/sq gpurun_out/$TAG/sq_summary.json
rm -f gpurun_out/$TAG/sq/*/b_kernel_trace.csv gpurun_out/$TAG/sq/*/*counter_collection.csv gpurun_out/$TAG/sq/b_kernel_trace.csv gpurun_out/$TAG/sq/*counter_collection.csv
