"""Same-process A/B of environment knobs on the DiT network evaluation alone (run on the GPU box).

    python tools/dit_ab.py [--batch 8] [--iters 20] [--prec bf16] "-" "VB_GEMM_P8_OFF=1" "VB_GEMM_PK=1" ...

Each argument is one environment setting ("-" = the default environment).  Every setting runs the same network evaluation (both CFG
branches, eager launches, no graph), is timed with HIP events over --iters evaluations, and its velocity / routes are compared bit for
bit with the first setting's.  Under rocprofv3 --kernel-trace --stats the kernels of the settings that differ appear as separate rows.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import _lib as L  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, DiTEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--prec", default="bf16")
ap.add_argument("--experts", type=int, default=4)
ap.add_argument("--T", type=int, default=752)
ap.add_argument("settings", nargs="*", default=["-"])
a = ap.parse_args()

device = torch.device("cuda:0")
B, T, Lc = a.batch, a.T, 80
dcfg = synth.DiTConfig(num_experts=a.experts)
sd = synth.make_state_dict(synth.dit_shapes(dcfg), 1234)
ctx = Context(device)
eng = DiTEngine(ctx, dcfg, sd, precision=a.prec)
inp = clip_batch(B, T, Lc)
cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"], inp["beats"], T)
t_idx = torch.full((2 * B,), 321, dtype=torch.int64)
lib = L.load()
ref = None
for s in a.settings:
    keys = []
    if s != "-":
        for kv in s.split():
            k, v = kv.split("=", 1)
            os.environ[k] = v
            keys.append(k)
    lib.vb_tune_reload()
    v, r = eng.forward(inp["x_latent"], t_idx, cond, seed=11, return_routes=True)
    torch.cuda.synchronize()
    v, r = v.clone(), r.clone()
    for _ in range(3):
        eng.forward(inp["x_latent"], t_idx, cond, seed=11)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        eng.forward(inp["x_latent"], t_idx, cond, seed=11)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    same = "reference" if ref is None else f"v equal={torch.equal(v, ref[0])} routes equal={torch.equal(r, ref[1])}"
    if ref is not None and not torch.equal(v, ref[0]):
        d = (v.double() - ref[0].double())
        same += f" rel_l2={float(d.norm() / ref[0].double().norm()):.3e} nan={int(torch.isnan(v).sum())}"
    print(f"[{s:40s}] {ms * 1e3:9.1f} us per evaluation (B={B}, {a.prec})   {same}", flush=True)
    if ref is None:
        ref = (v, r)
    for k in keys:
        del os.environ[k]
lib.vb_tune_reload()
