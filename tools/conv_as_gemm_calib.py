"""Calibration (GPU box): how fast does the DMA-fed GEMM run the VAE's wide convolutions if they were laid out as GEMMs
(rows = clip x time, K = taps x Ci, split precision = 3 bf16 passes)?  Compared with the conv kernel's time on the same layer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402

lib = L.load()
torch.manual_seed(0)
# (Ci, Co, T, k) of the VAE decoder's wide layers at 8 clips, and two vocoder layers
for Ci, Co, T, k in [(1536, 1536, 752, 5), (1536, 768, 1504, 5), (768, 768, 1504, 5), (768, 384, 1504, 5), (384, 384, 1504, 5), (256, 256, 12032, 3), (256, 256, 12032, 11),
                     (128, 128, 60160, 7)]:
    M, N, K = 8 * T, Co, k * Ci
    A = torch.randn(2, M, K, device="cuda").to(torch.bfloat16)
    Bw = (torch.randn(2, N, K, device="cuda") * 0.02).to(torch.bfloat16)
    Cd = torch.empty(M, N, device="cuda")
    for npl in (2, 1):
        for _ in range(2):
            L.check(lib.vb_gemm_bf16(L.ptr(A), L.ptr(Bw), None, M, N, K, npl, L.ptr(Cd), L.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            lib.vb_gemm_bf16(L.ptr(A), L.ptr(Bw), None, M, N, K, npl, L.ptr(Cd), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        passes = 3 if npl == 2 else 1
        print(f"Ci={Ci:5d} Co={Co:5d} T={T:6d} k={k:2d}  as GEMM {M}x{N}x{K} np={npl}: {us:8.1f} us  {2.0 * M * N * K * passes / us / 1e6:6.0f} TF/s (bf16 MFMA work)", flush=True)
