"""Print the per-pass kernel table from a rocprofv3 --stats csv (passes = warmup + timed steps of the profiled command)."""
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/trace/bench_kernel_stats.csv"
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = list(csv.DictReader(l for l in open(path) if not l.startswith("#")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms per pass {tot / passes / 1e6:.2f}")
for r in rows[:top]:
    print(f"{r['Name'][:70]:70s} calls/pass {int(r['Calls']) / passes:7.1f} avg {float(r['AverageNs']) / 1e3:8.1f} us  ms/pass {float(r['TotalDurationNs']) / passes / 1e6:7.2f}")
