#!/bin/bash
set -u
mkdir -p gpurun_out/r2m
export TMPDIR=/tmp
O=gpurun_out/r2m
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py tests/test_gpu_configs.py -m gpu -q -x --tb=short -p no:cacheprovider -k "conv or hifigan or vae or fullsize or bigvgan or melnet or overlap_add or cli" > $O/tests.log 2>&1
tail -5 $O/tests.log
python - <<'PY' 2>&1 | tee gpurun_out/r2m/conv_epi.log
import os, sys, torch
sys.path.insert(0, os.getcwd())
from versband_amd import _lib as L, pack
lib = L.load()
B = 8
cases = [(1536, 1536, 752, 3, 1, True, 0), (768, 768, 1504, 3, 1, True, 0), (384, 384, 1504, 3, 1, True, 0),
         (256, 256, 12032, 3, 1, True, 1), (256, 256, 12032, 11, 5, True, 1), (128, 128, 60160, 3, 1, True, 1), (128, 128, 60160, 7, 3, True, 1),
         (64, 64, 240640, 3, 1, True, 1), (32, 32, 481280, 3, 1, True, 1), (32, 1, 481280, 7, 1, False, 1)]
torch.manual_seed(0)
for Ci, Co, T, k, dil, res, act in cases:
    x = torch.randn(B, Ci, T, device="cuda"); w = torch.randn(Co, Ci, k, device="cuda") / (Ci * k) ** 0.5
    b = torch.randn(Co, device="cuda"); r = torch.randn(B, Co, T, device="cuda") if res else None
    out = torch.empty(B, Co, T, device="cuda"); wp = pack.pack_conv(w); wx3, cip = pack.pack_conv_x3(wp); pad = (k - 1) * dil // 2
    line = f"Ci={Ci:4d} Co={Co:4d} T={T:6d} k={k:2d} d={dil}:"
    outs = []
    for mode, env in (("direct", "1"), ("staged", None)):
        L.set_tuning(VB_CONV_DIRECT_EPI=env)
        def run():
            L.check(lib.vb_conv1d_f32(L.ptr(x), L.ptr(wp), L.ptr(b), B, Ci, T, Co, k, dil, pad, 1, 0, 0, T, act, 0.1,
                                      L.ptr(r) if res else None, L.ptr(out), L.ptr(wx3), cip, L.stream_ptr()), "conv")
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        outs.append(out.clone())
        line += f"  {mode} {e0.elapsed_time(e1) * 100:8.1f}us"
    line += "  equal=" + str(torch.equal(outs[0], outs[1]))
    print(line, flush=True)
PY
for e in 1 0; do
  if [ $e = 1 ]; then export VB_CONV_DIRECT_EPI=1; else unset VB_CONV_DIRECT_EPI; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > $O/bench_direct$e.json 2> $O/bench_direct$e.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_direct$e.json') if l.startswith('{')][-1])
print('direct_epi=$e', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'parity', d['parity_check'] and d['parity_check']['ok'], [ (c['class'][:14], round(c['ms_per_pass'],1)) for c in d['roofline']['classes']])
PY
done
