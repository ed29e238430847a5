#!/bin/bash
# One gpurun call: the driver's own bench command (stdout and stderr kept apart, the line checked), then the whole GPU suite.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-r06}
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_stdout.txt 2> gpurun_out/${tag}_bench_stderr.txt
echo "bench rc=$?"
python3 - <<PY
import json
t = open("gpurun_out/${tag}_bench_stdout.txt").read().strip().splitlines()
print("stdout lines:", len(t), "last line bytes:", len(t[-1]) if t else None)
d = json.loads(t[-1]); print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"], d.get("cpu_baseline"), d.get("split"), d["parity_check"])
PY
cp bench_detail.json gpurun_out/${tag}_bench_detail.json 2>/dev/null
if [ "${2:-}" != "nobench_only" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/${tag}_gpu_suite.log
  tail -8 gpurun_out/${tag}_gpu_suite.log
fi
