"""Timing-only ablations of the exact-fp32 convolution kernels on the vocoder's layer shapes (GPU box, EXPERIMENTS build:
VB_BUILD_EXPERIMENTS=1 python -m versband_amd.build).  Each column removes one thing from the kernel (results are wrong, only the
time means something): what is left says what the step time is made of.

    python tools/conv_f32_ablate.py [B, default 8]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import pack  # noqa: E402

lib = L.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PEAK = 157.3


def timed(run, n=6):
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


torch.manual_seed(0)
VAE_ONLY = len(sys.argv) > 2 and sys.argv[2] == "vae"
print("== conv1d_f32g_kernel<2,2,2,2,false,3> (128 x 128 tile): VB_F32G_ABL")
G_ABL = [("full", None), ("full", None), ("noFrag", 1), ("noDMA", 2), ("noBar", 4), ("noMFMA", 8), ("noEpi", 16), ("noFrag+DMA", 3), ("noDMA+Bar", 6), ("MFMA only", 7)]
for C, T, k, dil in (() if VAE_ONLY else ((256, 12032, 7, 3), (256, 12032, 3, 1), (128, 60160, 7, 3), (128, 60160, 11, 5))):
    x = torch.randn(B, C, T, device="cuda")
    w = torch.randn(C, C, k) / (C * k) ** 0.5
    wp = pack.pack_conv(w).cuda()
    b = torch.randn(C, device="cuda")
    r = torch.randn(B, C, T, device="cuda")
    out = torch.empty(B, C, T, device="cuda")
    pad = (k - 1) * dil // 2
    flops = 2.0 * B * C * C * k * T

    def run():
        L.check(lib.vb_conv1d_f32(L.ptr(x), L.ptr(wp), L.ptr(b), B, C, T, C, k, dil, pad, 1, 0, 0, T, 1, 0.1, L.ptr(r), L.ptr(out), None, 0,
                                  L.stream_ptr()), "conv")
    line = f"C={C:3d} T={T:6d} k={k:2d} d={dil} (ideal {flops / PEAK / 1e6:6.0f} us):"
    for name, v in G_ABL:
        if v is None:
            os.environ.pop("VB_F32G_ABL", None)
        else:
            os.environ["VB_F32G_ABL"] = str(v)
        line += f"  {name} {timed(run):6.0f}"
    os.environ.pop("VB_F32G_ABL", None)
    for tv in ("2", "3"):
        os.environ["VB_F32G_TILE"] = tv
        line += f"  tile{tv} {timed(run):6.0f}"
    os.environ.pop("VB_F32G_TILE", None)
    line += f"  full {timed(run):6.0f}"
    print(line, flush=True)

print("== conv1d_f32g_kernel<4,1,1,3,false,3> (128 x 96 tile, the VAE decoder's layers): VB_F32G_ABL (100 = the 128 x 128 tile on the same launch)")
V_ABL = [("full", None), ("full", None), ("noFrag", 1), ("noDMA", 2), ("noBar", 4), ("noMFMA", 8), ("noEpi", 16), ("noFrag+DMA", 3), ("MFMA only", 7), ("128x128", 100)]
for C, T, k, dil in ((1536, 752, 3, 1), (768, 1504, 3, 1), (1536, 752, 1, 1)):
    x = torch.randn(B, C, T, device="cuda")
    w = torch.randn(C, C, k) / (C * k) ** 0.5
    wp = pack.pack_conv(w).cuda()
    b = torch.randn(C, device="cuda")
    out = torch.empty(B, C, T, device="cuda")
    pad = (k - 1) * dil // 2
    flops = 2.0 * B * C * C * k * T

    def run():
        L.check(lib.vb_conv1d_f32(L.ptr(x), L.ptr(wp), L.ptr(b), B, C, T, C, k, dil, pad, 1, 0, 0, T, 0, 0.0, None, L.ptr(out), None, 0,
                                  L.stream_ptr()), "conv")
    line = f"C={C:4d} T={T:6d} k={k:2d} d={dil} (ideal {flops / PEAK / 1e6:6.0f} us):"
    for name, v in V_ABL:
        if v is None:
            os.environ.pop("VB_F32G_ABL", None)
        else:
            os.environ["VB_F32G_ABL"] = str(v)
        line += f"  {name} {timed(run):6.0f}"
    os.environ.pop("VB_F32G_ABL", None)
    print(line, flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "vae":
    sys.exit(0)

print("== respair_f32_kernel: VB_PAIRF_ABL")
P_ABL = [("full", None), ("noFrag", 1), ("noDMA", 2), ("noBar", 4), ("noMFMA", 8), ("noEpi", 16), ("noMid", 32), ("noEpi+Mid", 48), ("MFMA only", 7)]
for C, T, k, dil in ((64, 240640, 3, 1), (64, 240640, 7, 3), (64, 240640, 11, 5), (32, 481280, 3, 1), (32, 481280, 11, 5)):
    x = torch.randn(B, C, T, device="cuda")
    w1, w2 = torch.randn(C, C, k) / (C * k) ** 0.5, torch.randn(C, C, k) / (C * k) ** 0.5
    p1, p2 = pack.pack_conv(w1).cuda(), pack.pack_conv(w2).cuda()
    b1, b2 = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    out = torch.zeros(B, C, T, device="cuda")
    flops = 2.0 * 2.0 * B * C * C * k * T

    def run():
        L.check(lib.vb_respair_f32(L.ptr(x), L.ptr(p1), L.ptr(b1), L.ptr(p2), L.ptr(b2), B, C, T, k, dil, 0.1, 1.0, 0.0, L.ptr(out),
                                   L.stream_ptr()), "pair")
    line = f"C={C:3d} T={T:6d} k={k:2d} d={dil} (ideal {flops / PEAK / 1e6:6.0f} us):"
    for name, v in P_ABL:
        if v is None:
            os.environ.pop("VB_PAIRF_ABL", None)
        else:
            os.environ["VB_PAIRF_ABL"] = str(v)
        line += f"  {name} {timed(run):6.0f}"
    os.environ.pop("VB_PAIRF_ABL", None)
    print(line, flush=True)
