"""How long does a 128x128xK tile take alone on a CU vs paired (2 blocks/CU)?  N=768 -> 6 column tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import _lib as _vbL

lib = L.load()
N, K = 768, 768
torch.manual_seed(0)
# clocks: spin the GPU up for ~0.3 s before timing anything, and run the sweep twice (second pass is the one to read)
_w = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(200):
    _w @ _w
torch.cuda.synchronize()
for mt in (21, 42, 85, 94, 128, 170, 256, 21, 42, 64, 85, 94, 106, 128, 150, 170, 213, 256):
    M = 128 * mt
    A = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(1, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Cd = torch.empty(M, N, device="cuda")
    line = f"M={M:6d} tiles={mt * 6:5d} ({mt * 6 / 256:4.2f}/CU):"
    for abl, nm in ((0, "full"), (4, "noStore"), (1, "noDMA"), (5, "noLoop")):
        _vbL.set_tuning(VB_GEMM_ABLATE=str(abl))
        for _ in range(20):
            lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        line += f"  {nm}: {e0.elapsed_time(e1) * 1e3 / 100:6.1f}us"
    _vbL.set_tuning(VB_GEMM_ABLATE="0")
    print(line, flush=True)
