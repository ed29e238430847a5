"""Shared helpers for the parity tests (inputs, injected noise, error metrics)."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import ref_cpu
from versband_amd import synth

SEED = 1234


def rel_l2(a, b) -> float:
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b) -> float:
    return float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


def describe(name, got, ref) -> str:
    got = torch.as_tensor(got).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    d = (got - ref).abs()
    i = int(d.argmax())
    idx = np.unravel_index(i, tuple(ref.shape)) if ref.dim() else ()
    return (f"{name}: rel_l2={rel_l2(got, ref):.3e} max_abs={float(d.max()):.3e} at {tuple(int(v) for v in idx)} "
            f"got={float(got.flatten()[i]):.6g} ref={float(ref.flatten()[i]):.6g} |ref|max={float(ref.abs().max()):.4g} "
            f"nan={int(torch.isnan(got).sum())}")


def clip_batch(B, T, L, clip0=0, seed=SEED):
    clips = [synth.make_clip_inputs(seed, clip0 + b, T, L=L) for b in range(B)]
    return {k: torch.stack([c[k] for c in clips]) for k in clips[0]}


def exp_noise(B, T, E, nfe, depth, clip0=0, seed=SEED):
    """per block (E1[N,2], E2[N,E], E3[N,E]) Exp(1) draws, rows b-major - the oracle's noise format."""
    out = []
    for i in range(depth):
        parts = []
        for gate, w in ((0, 2), (1, E), (2, E)):
            parts.append(torch.from_numpy(np.concatenate(
                [synth.gumbel_exponentials(seed, clip0 + b, nfe, i, gate, T, w) for b in range(B)], 0)))
        out.append(tuple(parts))
    return out


def gumbel_arrays(noise_by_branch):
    """[[per-block (E1,E2,E3)] per branch] -> (g1 [depth,rows,2], g2, g3) Gumbel = -log(E) for the HIP path,
    rows = branch-major then b-major (cond rows first)."""
    depth = len(noise_by_branch[0])
    gs = []
    for j in range(3):
        per_block = []
        for i in range(depth):
            per_block.append(torch.cat([-(nb[i][j].log()) for nb in noise_by_branch], 0))
        gs.append(torch.stack(per_block))
    return tuple(gs)


def gumbel_arrays_steps(noise_steps):
    """list over steps of [branch][block](E1,E2,E3) -> arrays [step, depth, rows, w]."""
    per = [gumbel_arrays(nb) for nb in noise_steps]
    return tuple(torch.stack([p[j] for p in per]) for j in range(3))


def have_gpu() -> bool:
    return torch.cuda.is_available()


def check_digest(t, g, prefix, rtol):
    """Compare a tensor with the reference digest written by oracle/gen_golden.py:digest (sum, L2, max-abs, sampled elements)."""
    a = torch.as_tensor(t).detach().double().cpu().reshape(-1)
    assert a.numel() == int(g[prefix + "shape_numel"][0]), (a.numel(), int(g[prefix + "shape_numel"][0]))
    l2, mx = float(g[prefix + "l2"][0]), float(g[prefix + "maxabs"][0])
    assert abs(float(a.norm()) - l2) <= rtol * l2, (prefix, float(a.norm()), l2)
    assert abs(float(a.abs().max()) - mx) <= 10 * rtol * mx, (prefix, float(a.abs().max()), mx)
    assert abs(float(a.sum()) - float(g[prefix + "sum"][0])) <= rtol * l2 * (a.numel() ** 0.5), (prefix, float(a.sum()), float(g[prefix + "sum"][0]))
    idx = torch.from_numpy(g[prefix + "idx"])
    err = (a[idx] - torch.from_numpy(g[prefix + "val"])).abs().max()
    assert float(err) <= 10 * rtol * mx, (prefix, float(err), mx)


_LOAD_CHILD = r'''
import sys, time, torch
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
x = torch.randn(8, 256, 60000, device="cuda")
a @ a
torch.nn.functional.leaky_relu(x, 0.1)
torch.cuda.synchronize()
print("ready", flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20):
        a @ a
        torch.nn.functional.leaky_relu(x, 0.1)
    torch.cuda.synchronize()
'''


class beside_load:
    """`with beside_load(seconds):` runs the body beside a second GPU process that streams HBM and keeps the matrix pipe busy (~0.5 GB of
    device memory) - the condition under which round 5 found a counted DMA wait one piece short (profiles/r05_conv_tail_race.txt; two ranks
    sharing a GPU are exactly this).  A determinism test that ran WITHOUT the load proves nothing about that class of bug, so the test is
    SKIPPED (not passed) when the child does not report ready within `startup` seconds."""

    def __init__(self, seconds: float, startup: float = 240.0):
        self.seconds, self.startup, self.child = seconds, startup, None

    def __enter__(self):
        import select
        import subprocess
        import sys
        import pytest
        self.child = subprocess.Popen([sys.executable, "-c", _LOAD_CHILD, str(self.seconds)], stdout=subprocess.PIPE, text=True)
        ready, _, _ = select.select([self.child.stdout], [], [], self.startup)     # (a fresh box pages torch in for a minute or two)
        if not (ready and self.child.stdout.readline().strip() == "ready"):
            self.__exit__(None, None, None)
            pytest.skip(f"the load process did not start within {self.startup:.0f} s: determinism under load not exercised")
        return self

    def alive(self) -> bool:
        return self.child is not None and self.child.poll() is None

    def __exit__(self, *exc):
        if self.child is not None:
            self.child.kill()
            self.child.wait()
            self.child = None
        return False
