"""Micro-benchmark of the conv kernels on the real VAE / HiFi-GAN layer shapes (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import pack  # noqa: E402
from versband_amd import _lib as _vbL

lib = L.load()
B = 8
cases = [  # Ci, Co, T, k, dil, res, act
    (1536, 1536, 752, 3, 1, True, 0), (768, 768, 1504, 3, 1, True, 0), (384, 384, 1504, 3, 1, True, 0), (384, 80, 1504, 5, 1, False, 0),
    (256, 256, 12032, 3, 1, True, 1), (256, 256, 12032, 11, 5, True, 1), (128, 128, 60160, 7, 3, True, 1),
    (64, 64, 240640, 3, 1, True, 1), (64, 64, 240640, 11, 5, True, 1), (32, 32, 481280, 3, 1, True, 1), (32, 32, 481280, 7, 3, True, 1),
    (32, 1, 481280, 7, 1, False, 1), (20, 768, 752, 5, 1, False, 0),
]
torch.manual_seed(0)
for Ci, Co, T, k, dil, res, act in cases:
    x = torch.randn(B, Ci, T, device="cuda")
    w = torch.randn(Co, Ci, k, device="cuda") / (Ci * k) ** 0.5
    b = torch.randn(Co, device="cuda")
    r = torch.randn(B, Co, T, device="cuda") if res else None
    out = torch.empty(B, Co, T, device="cuda")
    wp = pack.pack_conv(w)
    wx3, cip = pack.pack_conv_x3(wp)
    pad = (k - 1) * dil // 2
    line = f"Ci={Ci:4d} Co={Co:4d} T={T:6d} k={k:2d} d={dil}:"
    flops = 2.0 * B * Co * Ci * k * T
    byts = 4.0 * B * T * (Ci + Co * (2 if res else 1))
    for split, cfgv in ((False, 0), (True, 0), (True, 1), (True, 2), (True, 3), (True, 4)):
        _vbL.set_tuning(VB_CONV_ABLATE=str(cfgv))

        def run():
            L.check(lib.vb_conv1d_f32(L.ptr(x), L.ptr(wp), L.ptr(b), B, Ci, T, Co, k, dil, pad, 1, 0, 0, T, act, 0.1,
                                      L.ptr(r) if res else None, L.ptr(out), L.ptr(wx3) if split else None, cip, L.stream_ptr()), "conv")
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        line += (f"  f32 {us:7.1f}us" if not split else (f"  x3 {us:7.1f}us {flops / us / 1e6:6.1f}TF" if cfgv == 0 else
                 f"  {['', 'noX', 'noW', 'noMFMA', 'noEpi'][cfgv]} {us:7.1f}"))
    print(line, flush=True)
