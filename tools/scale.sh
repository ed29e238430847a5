#!/bin/bash
# 1 / 2 / 4 / 8-GPU scaling of the headline workload on ONE node (BASELINE configs[3]): one command for the first hardware run of the RCCL path.
#   bash tools/scale.sh [steps=5] [warmup=1] [gpus="1 2 4 8"]
# Every N runs `python bench.py --gpus N` (bench.py starts its own N ranks on 127.0.0.1, one per GPU, weights by ONE flat RCCL broadcast
# from rank 0, clips sharded by global clip index, no collective in the timed region) and prints its JSON line; the table at the end
# gives mel-s/s, the weak-scaling ratio against N = 1, the slowest / fastest rank and the broadcast rate.  Lines land in gpurun_out/scale/.
set -u
STEPS=${1:-5}; WARM=${2:-1}; GPUS=${3:-"1 2 4 8"}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
O=gpurun_out/scale; mkdir -p $O
HAVE=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
echo "visible GPUs: $HAVE"
for n in $GPUS; do
  if [ "$n" -gt "$HAVE" ]; then echo "N=$n: skipped ($HAVE GPUs visible)"; continue; fi
  timeout 1200 python bench.py --gpus $n --steps $STEPS --warmup $WARM --no-cpu-baseline --no-pmc --no-isolated --detail $O/n$n.detail.json > $O/n$n.json 2> $O/n$n.err
  echo "N=$n exit $?"; grep '^{' $O/n$n.json | tail -1
done
python - <<PY
import json, os
base = None
for n in "$GPUS".split():
    p = f"$O/n{n}.json"
    if not os.path.exists(p):
        continue
    ls = [l for l in open(p) if l.startswith("{")]
    if not ls:
        print(f"N={n}: no JSON line (see $O/n{n}.err)"); continue
    d = json.load(open(f"$O/n{n}.detail.json")); r = d["ranks"]      # the stdout line is the <= 4 KB summary; the side file has the rank table
    base = base or d["value"] / d["n_gpus"]
    pr = r["per_rank_ms"]
    print(f"N={d['n_gpus']}: {d['value']:9.1f} mel-s/s  {d['ms_per_step']:7.2f} ms/step  scaling {d['value'] / base:5.2f}x of N=1  ranks {min(pr):.1f}..{max(pr):.1f} ms"
          f"  parity {d['parity_check'] and d['parity_check'].get('ok')}  backend {r['backend']}  broadcast "
          + (f"{r['weight_broadcast_bytes'] / 1e6:.0f} MB in {r['weight_broadcast_ms']:.1f} ms = {r['weight_broadcast_gbps']:.1f} GB/s" if r.get('weight_broadcast_ms') else "n/a"))
PY
