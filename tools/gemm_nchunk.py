"""A/B of the column-chunked tile order of the 128x128 GEMM kernel (VB_GEMM_NCHUNK) on the wide-N DiT shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L
from versband_amd import _lib as _vbL
lib = L.load()
for M, N, K in ((12032, 2304, 768), (6016, 2304, 768)):
    A = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(1, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda")
    ref, line = None, f"{M}x{N}x{K}:"
    for c in (0, 6, 3, 9, 0, 6):
        _vbL.set_tuning(VB_GEMM_NCHUNK=str(c))
        for _ in range(3):
            L.check(lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(C), L.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
        assert torch.equal(ref, C), f"chunk {c} changes the result"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(C), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        line += f"  chunk={c}: {e0.elapsed_time(e1) * 1e3 / 40:6.1f}us"
    print(line, flush=True)
