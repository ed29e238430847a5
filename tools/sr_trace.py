"""Phase timeline of the fused score + router kernel (tuning): main loop / router epilogue per workgroup."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import _lib as L  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, DiTEngine  # noqa: E402

lib = L.load()
lib.vbdbg_sr_trace.argtypes = [C.c_void_p]
lib.vbdbg_sr_trace.restype = None
dev = torch.device("cuda:0")
dcfg = synth.DiTConfig()
eng = DiTEngine(Context(dev), dcfg, synth.make_state_dict(synth.dit_shapes(dcfg), 1234), precision="bf16")
B, T, Lc = 8, 752, 80
inp = clip_batch(B, T, Lc)
cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(dev), inp["midi"].to(dev), inp["beats"].to(dev), T)
t_idx = torch.full((2 * B,), 500, dtype=torch.int64)
x = inp["x_latent"].to(dev)
for _ in range(20):
    eng.forward(x, t_idx, cond, seed=3)
torch.cuda.synchronize()
tr = torch.zeros(4 * 4096, dtype=torch.int64, device=dev)
lib.vbdbg_sr_trace(L.ptr(tr))
eng.forward(x, t_idx, cond, seed=3)
torch.cuda.synchronize()
lib.vbdbg_sr_trace(None)
t = tr.cpu().view(-1, 4)
t = t[t[:, 3] != 0]
tick = 1.0 / 100.0          # s_memtime: 100 MHz constant clock
loop = (t[:, 1] - t[:, 0]).double() * tick
epi = (t[:, 2] - t[:, 1]).double() * tick
span = (t[:, 2].max() - t[:, 0].min()).double() * tick
print(f"{len(t)} workgroups; loop median {loop.median():.2f} us (max {loop.max():.2f}); router epilogue median {epi.median():.2f} (max {epi.max():.2f}); "
      f"first start to last end {span:.2f} us")
