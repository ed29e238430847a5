"""Does the routed SwiGLU GEMM pay for re-fetching its gathered A rows once per column tile (141 MB fetched against ~45 needed)?
Runs vb_grouped_swiglu at the bench shape (24 064 slots, 4 experts, D 768, H 512) with (mode 0) the real kind of gather - every token in
two groups - and (mode 1) every slot gathering one of 256 rows, so that A lives in every L2.  Compare the w1/w3 kernel's duration in
`rocprofv3 --kernel-trace --stats`.     python tools/gemm_l2_probe.py <mode> [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = L.load()
N, G, D, H = 12032, 4, 768, 512
g = torch.Generator().manual_seed(5)
e1 = torch.randint(0, 2, (N,), generator=g)          # one expert of the first pair, one of the second
e2 = 2 + torch.randint(0, 2, (N,), generator=g)
slots_g = torch.cat([e1, e2])
tok = torch.cat([torch.arange(N), torch.arange(N)])
order = torch.sort(slots_g, stable=True).indices
perm = tok[order].to(torch.int32)
cnt = torch.bincount(slots_g, minlength=G)
off = torch.zeros(G + 1, dtype=torch.int32)
off[1:] = torch.cumsum(cnt, 0)
if mode == 1:
    perm = (torch.arange(2 * N) % 256).to(torch.int32)
u = torch.randn(1, N, D).to(torch.bfloat16).cuda()
w13 = (torch.randn(1, G, 2 * H, D) * 0.03).to(torch.bfloat16).cuda()
w2 = (torch.randn(1, G, D, H) * 0.03).to(torch.bfloat16).cuda()
scale = torch.ones(2 * N).cuda()
hidden = torch.zeros(1, 2 * N, H, dtype=torch.bfloat16, device="cuda")
out = torch.zeros(N, D, device="cuda")
perm, off = perm.cuda(), off.cuda()
for _ in range(reps):
    L.check(lib.vb_grouped_swiglu(L.ptr(u), L.ptr(perm), L.ptr(off), G, 2 * N, L.ptr(w13), L.ptr(w2), L.ptr(scale), D, H, 1, L.ptr(hidden),
                                  L.ptr(out), L.stream_ptr()), "grouped_swiglu")
torch.cuda.synchronize()
print("mode", mode, "done")
