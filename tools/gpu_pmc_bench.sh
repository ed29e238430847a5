#!/bin/bash
set -u
mkdir -p gpurun_out/pmcb
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcb/$c -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcb/$c.log 2>&1
  echo "$c exit $?"
done
cd $R
python - <<'PY'
import csv, collections, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/pmcb/{c}/*counter_collection.csv")
    if not fs:
        print(c, "no csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:48]
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[c] = {k: {"launches": v[0], "sum_kb": v[1]} for k, v in agg.items()}
json.dump(out, open("gpurun_out/pmcb/summary.json", "w"), indent=1)
for c in out:
    print(c)
    for k, v in sorted(out[c].items(), key=lambda kv: -kv[1]["sum_kb"])[:14]:
        print(f"  {k:50s} n={v['launches']:6d} sum={v['sum_kb']/1e6:9.3f} GB  per-launch={v['sum_kb']/v['launches']/1e3:9.2f} MB")
PY
rm -f gpurun_out/pmcb/*/b_kernel_trace.csv
ls gpurun_out/pmcb/*
