"""The CFG sampler of 8 clips as 1 x 8, 2 x 4 and 4 x 2 concurrent sub-batches (one HIP stream + host thread + engine handle each): wall time.
    python tools/sampler_streams_probe.py [euler steps = 10]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import model as vm  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, DiTEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
device = torch.device("cuda:0")
T, Lc, TOTAL = 752, 80, 8
dcfg = synth.DiTConfig()
sd = synth.make_state_dict(synth.dit_shapes(dcfg), 1234)
ctx = Context(device)
idx, dts = vm.euler_tables(steps + 1)
first = None
ref = None
for S in (1, 2, 4):
    n = TOTAL // S
    ws = []
    for i in range(S):
        eng = DiTEngine(ctx, dcfg, sd, precision="bf16", share=first)
        first = first or eng
        inp = clip_batch(n, T, Lc, clip0=i * n)
        ws.append(dict(eng=eng, x0=inp["x_latent"].to(device), t5=torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(device), midi=inp["midi"].to(device),
                       beats=inp["beats"].to(device), stream=torch.cuda.Stream(device=device), clip0=i * n))

    def run(w, reps):
        torch.cuda.set_device(device)
        with torch.cuda.stream(w["stream"]):
            for k in range(reps):
                cond = w["eng"].precompute_cond(w["t5"], w["midi"], w["beats"], T, persistent=True)
                w["z"] = w["eng"].sample_cfg(w["x0"], cond, idx, dts, 3.0, seed=7, clip_base=w["clip0"])

    for w in ws:
        run(w, 3)
    torch.cuda.synchronize()
    z = torch.cat([w["z"] for w in ws])
    if ref is None:
        ref = z.clone()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        ths = [threading.Thread(target=run, args=(w, 2)) for w in ws]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 2)
    print(f"{S} x {n} clips: {best * 1e3:7.2f} ms per {steps}-step sampler call over 8 clips   (latents equal to the 1 x 8 run: {torch.equal(z, ref)})", flush=True)
