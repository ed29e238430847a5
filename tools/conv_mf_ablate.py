"""Timing-only ablations of conv1d_f32w_kernel (experiments build: VB_BUILD_EXPERIMENTS=1 python -m versband_amd.build; results are WRONG, only the
times mean something): VB_F32W_ABL bit 1 = no in-place window pass (LeakyReLU / padding), 2 = no epilogue, 4 = no window DMA after the first chunk.
    python tools/conv_mf_ablate.py [clips]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import pack  # noqa: E402

lib = L.load()
assert lib.vb_has_experiments(), "needs the experiments build"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
for Ci, Co, T, k, dil, act in ((256, 256, 12032, 3, 1, 1), (256, 256, 12032, 11, 1, 1), (128, 128, 60160, 3, 1, 1), (1536, 1536, 752, 3, 1, 0), (64, 64, 240640, 3, 1, 1)):
    x = torch.randn(B, Ci, T, device="cuda")
    w = torch.randn(Co, Ci, k) / (Ci * k) ** 0.5
    b = torch.randn(Co, device="cuda")
    r = torch.randn(B, Co, T, device="cuda")
    out = torch.empty(B, Co, T, device="cuda")
    wp, wm = pack.pack_conv(w).cuda(), pack.pack_conv_mf(w).cuda()
    pad = (k - 1) * dil // 2
    line = f"Ci={Ci:4d} T={T:6d} k={k:2d}:"
    for abl, name in ((0, "full"), (1, "no window pass"), (2, "no epilogue"), (4, "no window DMA"), (3, "no pass, no epilogue"), (7, "loop only")):
        os.environ["VB_F32W_ABL"] = str(abl)

        def run():
            L.check(lib.vb_conv1d_f32_mf(L.ptr(x), L.ptr(wp), L.ptr(wm), L.ptr(b), B, Ci, T, Co, k, dil, pad, T, act, 0.1, L.ptr(r), 1.0, 0.0, L.ptr(out),
                                         L.stream_ptr()), "mf")
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            run()
        e1.record()
        torch.cuda.synchronize()
        line += f"  {name} {e0.elapsed_time(e1) * 1e3 / 8:7.1f}"
    print(line, flush=True)
