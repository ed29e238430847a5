"""8-wave 256x256 ping-pong GEMM (VB_GEMM_P8) against the 4-wave kernels on the DiT shapes: bitwise-equal results, microseconds,
TFLOP/s.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402

lib = L.load()
shapes = [(256, 256, 768), (12032, 768, 768), (12032, 2304, 768), (12032, 1024, 768), (12032, 768, 512), (12032, 640, 768), (24064, 1024, 768),
          (6016, 768, 768), (6016, 2304, 768), (48128, 2304, 768), (3008, 2304, 768), (1504, 2304, 768)]
torch.manual_seed(0)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M, N, K in shapes:
    A = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(1, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    Cd = torch.empty(M, N, device="cuda")

    def run():
        L.check(lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), L.ptr(bias), M, N, K, 1, L.ptr(Cd), L.stream_ptr()), "gemm")
    line = f"{M:6d}x{N:5d}x{K:4d}:"
    L.set_tuning(VB_GEMM_P8="0")
    us = timed(run)
    ref = Cd.clone()
    line += f"  4-wave {us:7.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF"
    for v, nm in ((1, "pp32x5"), (3, "pp64x2"), (4, "sw32x5"), (5, "sw32x4"), (6, "sw64x2")):
        L.set_tuning(VB_GEMM_P8=str(v))
        Cd.fill_(float("nan"))
        us = timed(run)
        same = torch.equal(ref, Cd)
        line += f" | {nm} {us:7.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF {'==' if same else 'DIFF ' + format(float((ref - Cd).abs().max()), '.2e')}"
    for v, nm in ((11, "noDMA"), (12, "noLDSrd"), (13, "noMFMA"), (14, "noBarrier"), (15, "noEpi")):
        L.set_tuning(VB_GEMM_P8=str(v))
        line += f" | {nm} {timed(run):6.1f}"
    L.set_tuning(VB_GEMM_P8=None)
    print(line, flush=True)
# split precision (3 segments) on one shape
M, N, K = 12032, 768, 768
A = torch.randn(2, M, K, device="cuda").to(torch.bfloat16)
B = (torch.randn(2, N, K, device="cuda") * 0.05).to(torch.bfloat16)
Cd = torch.empty(M, N, device="cuda")
outs = []
for v in ("0", "1"):
    L.set_tuning(VB_GEMM_P8=v)
    us = timed(lambda: L.check(lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 2, L.ptr(Cd), L.stream_ptr()), "gemm"))
    outs.append((Cd.clone(), us))
L.set_tuning(VB_GEMM_P8=None)
print(f"split (bf16x3) {M}x{N}x{K}: 4-wave {outs[0][1]:.1f}us, p8 {outs[1][1]:.1f}us, equal={torch.equal(outs[0][0], outs[1][0])}")
