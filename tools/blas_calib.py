"""Calibration only (not a product path): what the vendor bf16 GEMM (torch.mm -> hipBLASLt/rocBLAS) takes on the DiT's shapes."""
import torch

torch.manual_seed(0)
for M, N, K in ((12032, 768, 768), (12032, 2304, 768), (24064, 1024, 768), (12032, 768, 512), (12032, 1024, 192), (12032, 192, 512)):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"M={M} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
