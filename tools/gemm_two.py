"""Two DiT GEMM shapes, a few launches each (PMC passes: tools/gpu_pmc_l2.sh style, kept tiny)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L
lib = L.load()
for M, N, K in ((12032, 768, 768), (12032, 2304, 768)):
    A = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(1, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda")
    for _ in range(4):
        L.check(lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(C), L.stream_ptr()), "gemm")
    torch.cuda.synchronize()
