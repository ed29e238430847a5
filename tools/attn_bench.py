"""Micro-benchmark + ablation of the attention kernel at the bench shape (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import _lib as _vbL

lib = L.load()
B, H, hd, Lc = 16, 8, 96, 80
for T in (752, 376):
    Tpad, Lpad = (T + 63) // 64 * 64, 128
    mk = lambda *s: (torch.randn(*s, device="cuda")).to(torch.bfloat16)  # noqa: E731
    q, k, vt = mk(1, B, T, H, hd), mk(1, B, T, H, hd), mk(1, B, H, hd, Tpad)
    ky, vyt = mk(1, B, Lc, H, hd), mk(1, B, H, hd, Lpad)
    cw = torch.randn(H, device="cuda")
    out = torch.empty(1, B, T, H, hd, dtype=torch.bfloat16, device="cuda")
    for name, use_self in (("self+cross", True), ("cross-only", False)):
        line = f"T={T} {name:10s}:"
        for abl, nm in ((0, "full"), (1, "noKVload"), (2, "noSoftmax"), (3, "noPV"), (4, "noQK")):
            _vbL.set_tuning(VB_ATTN_ABLATE=str(abl))

            def run():
                L.check(lib.vb_attention(L.ptr(q), L.ptr(k) if use_self else None, L.ptr(vt) if use_self else None, L.ptr(ky), L.ptr(vyt),
                                         L.ptr(cw), B, T, Tpad, Lc, Lpad, H, hd, 1, L.ptr(out), L.stream_ptr()), "attn")
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            fl = 4.0 * B * H * T * hd * ((T if use_self else 0) + Lc)
            line += f"  {nm}: {us:6.1f}us" + (f" ({fl / us / 1e6:4.0f}TF)" if abl == 0 else "")
        print(line, flush=True)
_vbL.set_tuning(VB_ATTN_ABLATE="0")
