"""Per-block timeline of the bf16 GEMM (tuning tool): which CU ran each 128x128 tile, when, for how long."""
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402

lib = L.load()
lib.vbdbg_gemm_trace.argtypes = [C.c_void_p]
lib.vbdbg_gemm_trace.restype = None
N, K = 768, 768
_w = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(200):
    _w @ _w
torch.cuda.synchronize()
for M in [int(a) for a in (sys.argv[1:] or ["12032", "10880", "16384"])]:
    A = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(1, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Cd = torch.empty(M, N, device="cuda")
    nblk = 6 * ((M + 127) // 128 + 8) // 8 * 8 + 64
    tr = torch.zeros(nblk * 4 + 64, dtype=torch.int64, device="cuda")
    for _ in range(20):
        lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr())
    torch.cuda.synchronize()
    lib.vbdbg_gemm_trace(L.ptr(tr))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), None, M, N, K, 1, L.ptr(Cd), L.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    lib.vbdbg_gemm_trace(None)
    t = tr.cpu().view(-1, 4)[:nblk]
    t = t[t[:, 2] != 0]
    t0 = int(t[:, 0].min())
    span = int(t[:, 2].max()) - t0
    ev_us = e0.elapsed_time(e1) * 1e3
    print(f"ABL={os.environ.get('VB_GEMM_ABLATE', 0)} M={M}: {len(t)} blocks traced, span {span} ticks, event {ev_us:.1f} us (incl. launch) ")
    tick = 1.0 / 2100.0   # s_memtime ticks at about the shader clock here (calibrated against event times); per-XCC bases differ
    st = (t[:, 0] - t0).double() * tick
    lp = (t[:, 1] - t[:, 0]).double() * tick
    ep = (t[:, 2] - t[:, 1]).double() * tick
    print(f"  loop us:   min {lp.min():.2f} median {lp.median():.2f} p90 {lp.quantile(0.9):.2f} max {lp.max():.2f}")
    print(f"  epilogue:  min {ep.min():.2f} median {ep.median():.2f} p90 {ep.quantile(0.9):.2f} max {ep.max():.2f}")
    hw = t[:, 3]
    pro = ((hw >> 36) & 0xFFFFFFF).double() * tick
    if pro.max() > 0:
        print(f"  prologue:  min {pro.min():.2f} median {pro.median():.2f} p90 {pro.quantile(0.9):.2f} max {pro.max():.2f}   (inside 'loop')")
    xcc = (hw >> 32) & 0xF
    hwid = hw & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    key = (xcc * 64 + se * 16 + sh * 8).long() * 16 + cu
    per = collections.Counter(key.tolist())
    print(f"  distinct CUs {len(per)}; blocks/CU histogram {sorted(collections.Counter(per.values()).items())}; per XCC {sorted(collections.Counter(xcc.tolist()).items())}")
    # concurrency per CU at the time each block starts
    # first 3 CUs: list their blocks
    for k in list(per)[:3]:
        idx = [i for i in range(len(t)) if int(key[i]) == k]
        b0 = min(float(st[i]) for i in idx)
        rows = [(round(float(st[i]) - b0, 2), round(float(st[i] + lp[i]) - b0, 2), round(float(st[i] + lp[i] + ep[i]) - b0, 2)) for i in idx]
        print(f"    CU key {k}: {rows}")
