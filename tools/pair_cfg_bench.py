"""Ring shapes of the fused fp32 ResBlock pair at 64 / 32 channels (GPU box, EXPERIMENTS build): VB_PAIRF_CFG=1 / 2 = one tap per ring step
with 4 / 3 stages at 64 channels (53 / 49 KB of LDS: three workgroups per CU instead of two), 3 = two taps / 4 stages at 32 channels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import pack  # noqa: E402

lib = L.load()
B = 8
torch.manual_seed(0)


def timed(run, n=8):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for C, T, cfgs in ((64, 240640, (None, 5, 1, None, 5, 1)), (32, 481280, (None, 4, 3, None, 4, 3))):
    for k, dil in ((3, 1), (7, 3), (11, 5)):
        x = torch.randn(B, C, T, device="cuda")
        p1, p2 = pack.pack_conv(torch.randn(C, C, k) / (C * k) ** 0.5).cuda(), pack.pack_conv(torch.randn(C, C, k) / (C * k) ** 0.5).cuda()
        b1, b2 = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        out = torch.zeros(B, C, T, device="cuda")

        def run():
            L.check(lib.vb_respair_f32(L.ptr(x), L.ptr(p1), L.ptr(b1), L.ptr(p2), L.ptr(b2), B, C, T, k, dil, 0.1, 1.0, 0.0, L.ptr(out),
                                       L.stream_ptr()), "pair")
        timed(run)
        line = f"C={C} k={k:2d}:"
        for v in cfgs:
            if v is None:
                os.environ.pop("VB_PAIRF_CFG", None)
            else:
                os.environ["VB_PAIRF_CFG"] = str(v)
            line += f"  cfg {v if v else 'product'}: {timed(run):6.0f} us"
        os.environ.pop("VB_PAIRF_CFG", None)
        print(line, flush=True)
