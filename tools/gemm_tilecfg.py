"""bf16 GEMM tile configurations (VB_GEMM_TILE) on the DiT's shapes: time + bit-equality against the 128x128 kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import _lib as _vbL

lib = L.load()
_w = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(200):
    _w @ _w
torch.cuda.synchronize()
shapes = [(12032, 768, 768), (12032, 2304, 768), (24064, 1024, 768), (12032, 768, 512), (12032, 1024, 192), (6016, 768, 768), (6016, 2304, 768)]
cfgs = ("22", "33", "")
if len(sys.argv) > 1 and sys.argv[1] == "small":       # one / two / four clips (x 2 CFG branches): the small-tile configurations
    shapes = [(1504, 768, 768), (1504, 2304, 768), (1504, 1024, 768), (1504, 768, 512), (3008, 768, 768), (3008, 2304, 768),
              (3008, 768, 512), (6016, 768, 768), (6016, 2304, 768)]
    cfgs = ("22", "11", "21", "33", "")
torch.manual_seed(0)
for M, N, K in shapes:
    A = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(1, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Cd = torch.empty(M, N, device="cuda")
    ref = None
    line = f"{M:6d}x{N:5d}x{K:4d}:"
    for cfg in cfgs:
        if cfg:
            _vbL.set_tuning(VB_GEMM_TILE=cfg)
        else:
            _vbL.set_tuning(VB_GEMM_TILE=None)
        Cd.zero_()
        for _ in range(10):
            L.check(lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        if ref is None:
            ref = Cd.clone()
            exact = (A[0].float() @ B[0].float().t())
            err = ((ref - exact).abs().max() / exact.abs().max()).item()
            line += f" (err vs fp32 {err:.1e})"
        ok = torch.equal(ref, Cd)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        e0.record()
        for _ in range(n):
            lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        line += f"  {cfg or 'auto'}: {us:6.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF{'' if ok else ' MISMATCH'}"
    print(line, flush=True)
