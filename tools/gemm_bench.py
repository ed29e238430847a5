"""Micro-benchmark of the bf16 GEMM variants (VB_GEMM_VARIANT) on the DiT shapes; run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import _lib as _vbL

lib = L.load()
shapes = [(12032, 768, 768), (12032, 2304, 768), (12032, 1024, 768), (12032, 768, 512), (24064, 1024, 768), (48128, 768, 768)]
torch.manual_seed(0)
for M, N, K in shapes:
    A = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(1, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Cd = torch.empty(M, N, device="cuda")
    ref = None
    line = f"{M:6d}x{N:5d}x{K:4d}:"
    for variant in (0, 1):
        _vbL.set_tuning(VB_GEMM_VARIANT=str(variant))
        for _ in range(3):
            L.check(lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        if ref is None:
            ref = Cd.clone()
        else:
            assert torch.equal(ref, Cd), f"variant {variant} differs from variant 0"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        line += f"  v{variant}: {us:7.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF"
    _vbL.set_tuning(VB_GEMM_VARIANT="1")
    for abl, nm in ((1, "noDMA"), (3, "noLDSread"), (4, "noStore"), (5, "noLoop")):
        _vbL.set_tuning(VB_GEMM_ABLATE=str(abl))
        for _ in range(2):
            lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        line += f"  {nm}: {e0.elapsed_time(e1) * 1e3 / 30:6.1f}us"
    _vbL.set_tuning(VB_GEMM_ABLATE="0")
    print(line, flush=True)
