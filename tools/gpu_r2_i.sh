#!/bin/bash
# per-kernel times (rocprofv3 kernel stats, one stream, eager) with the 8-wave kernel enabled per epilogue
set -u
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2i
cd /tmp
prof() { # name env
  env $2 VB_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -o b -- python $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-isolated > $O/$1.log 2>&1
  f=$(find $O/$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1 ($2)"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "gemm_bf16" in n or "band_ffn" in n:
        print(f"  {n[:60]:60s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
  find $O/$1 -name "*kernel_trace.csv" -delete
}
prof base "A=1"
prof qkv_staged "VB_GEMM_P8_MASK=4"
prof qkv_direct "VB_GEMM_P8_MASK=4 VB_GEMM_P8_DIRECT=4"
prof swiglu_staged "VB_GEMM_P8_MASK=16"
prof swiglu_direct "VB_GEMM_P8_MASK=16 VB_GEMM_P8_DIRECT=16"
cd $R
