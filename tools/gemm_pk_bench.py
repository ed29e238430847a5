"""Persistent 8-wave GEMM (gemm_bf16_pk_kernel) on the DiT shapes, fp32 output, against the per-tile 8-wave and the 4-wave kernels:
bitwise-equal results, microseconds, TFLOP/s, timing-only ablations and a per-stage s_memtime trace.  Experiments build
(VB_BUILD_EXPERIMENTS=1 python -m versband_amd.build); run on the GPU box.

    python tools/gemm_pk_bench.py [trace]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402

lib = L.load()
lib.vbdbg_gemm_trace.argtypes = [C.c_void_p]
lib.vbdbg_gemm_trace.restype = None
shapes = [(12032, 2304, 768), (12032, 768, 768), (6016, 2304, 768), (24064, 1024, 768), (12032, 768, 512), (1504, 2304, 768)]
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


ABL = ((2, "noDMA"), (3, "noLDSrd"), (4, "noMFMA"), (6, "noEpi"), (7, "noDMA+rd"), (8, "DMA only"))
for M, N, K in shapes:
    A = torch.randn(1, M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(1, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    Cd = torch.empty(M, N, device="cuda")

    def run():
        L.check(lib.vb_gemm_bf16(L.ptr(A), L.ptr(B), L.ptr(bias), M, N, K, 1, L.ptr(Cd), L.stream_ptr()), "gemm")
    fl = 2.0 * M * N * K
    line = f"{M:6d}x{N:5d}x{K:4d}:"
    L.set_tuning(VB_GEMM_P8="0", VB_GEMM_PK_F32=None)
    us = timed(run)
    ref = Cd.clone()
    line += f"  4-wave {us:6.1f}us {fl / us / 1e6:4.0f}TF"
    L.set_tuning(VB_GEMM_P8="4")
    us = timed(run)
    line += f" | p8 sw32x5 {us:6.1f}us {fl / us / 1e6:4.0f}TF {'==' if torch.equal(ref, Cd) else 'DIFF'}"
    L.set_tuning(VB_GEMM_P8="15")
    line += f" (noEpi {timed(run):5.1f})"
    L.set_tuning(VB_GEMM_P8=None, VB_GEMM_PK_F32="1")
    Cd.fill_(float("nan"))
    us = timed(run)
    line += f" | pk {us:6.1f}us {fl / us / 1e6:4.0f}TF {'==' if torch.equal(ref, Cd) else 'DIFF ' + format(float((ref - Cd).abs().max()), '.2e')}"
    for code, nm in ABL:
        L.set_tuning(VB_GEMM_PK_F32=str(code))
        line += f" | {nm} {timed(run):5.1f}"
    print(line, flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        L.set_tuning(VB_GEMM_PK_F32="100")
        tr = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        lib.vbdbg_gemm_trace(L.ptr(tr))
        run()
        torch.cuda.synchronize()
        lib.vbdbg_gemm_trace(None)
        t = tr.cpu().view(256, 64)
        KT = K // 64
        # layout per workgroup: [0] start, then per tile: KT x (arrive, release), tile end
        rows = []
        for b in range(256):
            r = t[b]
            n = int((r != 0).sum())
            if n < 2 + 2 * KT:
                continue
            i, tiles = 1, []
            while i + 2 * KT < n:
                st = r[i:i + 2 * KT].view(KT, 2)
                end = int(r[i + 2 * KT])
                tiles.append((st, end))
                i += 2 * KT + 1
            rows.append((b, int(r[0]), tiles))
        import statistics as S
        waits = [int(st[k, 1] - st[k, 0]) for _, _, tiles in rows for st, _ in tiles for k in range(KT)]
        comps = [int(st[k + 1, 0] - st[k, 1]) for _, _, tiles in rows for st, _ in tiles for k in range(KT - 1)]
        epis = [int(end - st[KT - 1, 1]) for _, _, tiles in rows for st, end in tiles]
        first = [int(tiles[0][0][0, 0] - t0) for _, t0, tiles in rows]
        total = [int(tiles[-1][1] - t0) for _, t0, tiles in rows]
        q = lambda v, f: sorted(v)[int(f * (len(v) - 1))]
        print(f"    trace: {len(rows)} workgroups, tiles per workgroup {S.mean(len(tl) for _, _, tl in rows):.2f}; ticks (s_memtime = shader cycles):"
              f" stage-end wait median {S.median(waits)} p90 {q(waits, 0.9)}; multiply span between stage ends median {S.median(comps)} p90 {q(comps, 0.9)};"
              f" last stage end -> tile end (4th k-step + epilogue) median {S.median(epis)} p90 {q(epis, 0.9)}; start -> first stage end median {S.median(first)};"
              f" whole workgroup median {S.median(total)} max {max(total)}")
        b, t0, tiles = rows[len(rows) // 2]
        print(f"    workgroup {b}: " + " | ".join(" ".join(f"{int(st[k, 0] - t0)}+{int(st[k, 1] - st[k, 0])}" for k in range(KT)) + f" end {end - t0}" for st, end in tiles))
    L.set_tuning(VB_GEMM_PK_F32=None)
