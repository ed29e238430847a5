#!/bin/bash
# L2 attribution of the GEMM launches (round-2 experiment): hit rate and fabric reads per launch of the GEMM
# micro-benchmark.  One counter group per pass (TCC has 4 counter slots; FETCH_SIZE alone costs 3), --kernel-trace only.
set -u
mkdir -p gpurun_out/pmc_l2
export TMPDIR=/tmp
R=$PWD
cd /tmp
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $grp | tr ' ' '+')
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_l2/$tag -o g -- python $R/tools/gemm_bench.py > $R/gpurun_out/pmc_l2/$tag.log 2>&1
  echo "$tag exit $?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_l2/*/*counter_collection*.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    print(f)
    for (kn, cn), (n, v) in sorted(agg.items()):
        print(f"  {kn:60s} {cn:22s} launches={n:5d} per_launch={v / n:14.1f}")
PY
