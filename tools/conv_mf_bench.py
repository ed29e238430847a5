"""fp32 minimal filtering (conv1d_f32w_kernel) against the direct fp32 kernel (conv1d_f32g_kernel) on the vocoder's unfused ResBlock layer shapes
and the VAE's 3-tap layers (run on the GPU box): time per launch, the ratio, and the largest difference in units of the output's max-abs.
    python tools/conv_mf_bench.py [clips]        VB_MF_OCC=2 selects the two-workgroups-per-CU build"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import pack  # noqa: E402

lib = L.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cases = [  # Ci, Co, T, k, dil, res, act
    (256, 256, 12032, 3, 1, True, 1), (256, 256, 12032, 3, 3, False, 1), (256, 256, 12032, 3, 5, False, 1),
    (256, 256, 12032, 7, 1, True, 1), (256, 256, 12032, 7, 3, False, 1), (256, 256, 12032, 7, 5, False, 1),
    (256, 256, 12032, 11, 1, True, 1), (256, 256, 12032, 11, 3, False, 1), (256, 256, 12032, 11, 5, False, 1),
    (128, 128, 60160, 3, 1, True, 1), (128, 128, 60160, 7, 3, False, 1), (128, 128, 60160, 11, 5, False, 1), (128, 128, 60160, 11, 1, True, 1),
    (64, 64, 240640, 7, 3, False, 1), (64, 64, 240640, 11, 1, True, 1),
    (1536, 1536, 752, 3, 1, True, 0), (768, 768, 1504, 3, 1, True, 0), (384, 384, 1504, 3, 1, True, 0),
]
torch.manual_seed(0)


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot_d = tot_m = 0.0
for Ci, Co, T, k, dil, res, act in cases:
    x = torch.randn(B, Ci, T, device="cuda")
    w = torch.randn(Co, Ci, k) / (Ci * k) ** 0.5
    b = torch.randn(Co, device="cuda")
    r = torch.randn(B, Co, T, device="cuda") if res else None
    od, om = torch.empty(B, Co, T, device="cuda"), torch.empty(B, Co, T, device="cuda")
    wp, wm = pack.pack_conv(w).cuda(), pack.pack_conv_mf(w).cuda()
    pad = (k - 1) * dil // 2

    def direct():
        L.check(lib.vb_conv1d_f32(L.ptr(x), L.ptr(wp), L.ptr(b), B, Ci, T, Co, k, dil, pad, 1, 0, 0, T, act, 0.1, L.ptr(r) if res else None,
                                  L.ptr(od), None, 0, L.stream_ptr()), "conv")

    def mf():
        L.check(lib.vb_conv1d_f32_mf(L.ptr(x), L.ptr(wp), L.ptr(wm), L.ptr(b), B, Ci, T, Co, k, dil, pad, T, act, 0.1, L.ptr(r) if res else None,
                                     1.0, 0.0, L.ptr(om), L.stream_ptr()), "conv mf")

    td, tm = timed(direct), timed(mf)
    err = float((od - om).abs().max() / od.abs().max())
    tot_d += td
    tot_m += tm
    flops = 2.0 * B * Co * Ci * k * T
    print(f"Ci={Ci:4d} Co={Co:4d} T={T:6d} k={k:2d} d={dil}: direct {td:8.1f} us ({flops / td / 1e6:6.1f} TF/s)   mf {tm:8.1f} us "
          f"({flops * pack.mf_pseudo_taps(k) / (2 * k) / tm / 1e6:6.1f} TF/s executed, {flops / tm / 1e6:6.1f} direct-equivalent)   "
          f"x{td / tm:5.2f}   max|d|/max|y| {err:.2e}", flush=True)
print(f"sum: direct {tot_d:.0f} us, mf {tot_m:.0f} us, x{tot_d / tot_m:.3f}")
# fused 32- / 64-channel ResBlock pairs: respair_f32_kernel (direct) against respair_f32w_kernel (minimal filtering)
for C, T, k, dil in [(32, 481280, k, d) for k, d in ((3, 1), (3, 3), (7, 1), (7, 3), (11, 1), (11, 5))] + \
                    [(64, 240640, k, d) for k, d in ((3, 1), (3, 3), (7, 1), (7, 3), (11, 1), (11, 5))]:
    x = torch.randn(B, C, T, device="cuda")
    w1, w2 = torch.randn(C, C, k) / (C * k) ** 0.5, torch.randn(C, C, k) / (C * k) ** 0.5
    b1, b2 = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    p1, p2, m1, m2 = pack.pack_conv(w1).cuda(), pack.pack_conv(w2).cuda(), pack.pack_conv_mf(w1).cuda(), pack.pack_conv_mf(w2).cuda()
    od, om = torch.empty(B, C, T, device="cuda"), torch.empty(B, C, T, device="cuda")

    def pd():
        L.check(lib.vb_respair_f32(L.ptr(x), L.ptr(p1), L.ptr(b1), L.ptr(p2), L.ptr(b2), B, C, T, k, dil, 0.1, 1.0, 0.0, L.ptr(od), L.stream_ptr()), "pair")

    def pm():
        L.check(lib.vb_respair_f32_mf(L.ptr(x), L.ptr(m1), L.ptr(b1), L.ptr(m2), L.ptr(b2), B, C, T, k, dil, 0.1, 1.0, 0.0, L.ptr(om), L.stream_ptr()), "pair mf")

    td, tm = timed(pd, 5), timed(pm, 5)
    flops = 2.0 * 2.0 * B * C * C * k * T
    print(f"pair C={C} T={T} k={k:2d} d={dil}: direct {td:8.1f} us ({flops / td / 1e6:6.1f} TF/s)   mf {tm:8.1f} us ({flops * pack.mf_pseudo_taps(k) / (2 * k) / tm / 1e6:6.1f} TF/s "
          f"executed)   x{td / tm:5.2f}   max|d|/max|y| {float((od - om).abs().max() / od.abs().max()):.2e}", flush=True)
