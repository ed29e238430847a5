"""Are the GEMM tile configurations bit-identical through a whole DiT forward (all epilogues)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, DiTEngine  # noqa: E402
from versband_amd import _lib as _vbL

dev = torch.device("cuda:0")
dcfg = synth.DiTConfig()
sd = synth.make_state_dict(synth.dit_shapes(dcfg), 1234)
ctx = Context(dev)
eng = DiTEngine(ctx, dcfg, sd, precision="bf16")
B, T, Lc = 4, 752, 80
inp = clip_batch(B, T, Lc)
t5 = torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(dev)
cond = eng.precompute_cond(t5, inp["midi"].to(dev), inp["beats"].to(dev), T)
t_idx = torch.full((2 * B,), 500, dtype=torch.int64)
outs = {}
for cfg in ("22", "33"):
    _vbL.set_tuning(VB_GEMM_TILE=cfg)
    v, r = eng.forward(inp["x_latent"].to(dev), t_idx, cond, seed=3, return_routes=True)
    torch.cuda.synchronize()
    outs[cfg] = (v.clone(), r.clone())
v0, r0 = outs["22"]
v1, r1 = outs["33"]
print("routes equal:", torch.equal(r0, r1), "flips", int((r0 != r1).sum()), "of", r0.numel())
print("v equal:", torch.equal(v0, v1), "max abs diff", float((v0 - v1).abs().max()), "rel", float((v0 - v1).norm() / v0.norm()))
