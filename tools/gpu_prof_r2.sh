#!/bin/bash
# round-2 profiles: rocprofv3 kernel stats of the default bench command, the one-stream command and batch 1; PMC FETCH / WRITE passes
set -u
TAG=${1:-r2p}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
cd /tmp
prof() { # name args
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-pmc --no-parity-check $2 > $O/$1.log 2>&1
  f=$(find $O/$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${1}_kernel_stats.csv
  tail -2 $O/$1.log | cut -c1-300
  python $R/tools/prof_summary.py $O/${1}_kernel_stats.csv 2 14
  find $O/$1 -name "*kernel_trace.csv" -delete
}
prof default ""
prof stream1 "--streams 1"
prof batch1 "--streams 1 --batch 1"
if [ "${2:-}" = "pmc" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-pmc --no-parity-check --streams 1 > $O/$c.log 2>&1
  echo "$c exit $?"
done
cd $R
python - <<PY
import csv, collections, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/$TAG/{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(c, "no csv"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[c] = {k: {"launches": v[0], "sum_kb": v[1]} for k, v in agg.items()}
json.dump(out, open("gpurun_out/$TAG/pmc_summary.json", "w"), indent=1)
for c in out:
    print(c)
    for k, v in sorted(out[c].items(), key=lambda kv: -kv[1]["sum_kb"])[:16]:
        print(f"  {k:60s} n={v['launches']:6d} sum={v['sum_kb']/1e6:9.3f} GB  per-launch={v['sum_kb']/v['launches']/1e3:9.2f} MB")
PY
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
find gpurun_out/$TAG -name "*counter_collection.csv" -size +8M -delete
fi
cd $R
