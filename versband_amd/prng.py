"""Portable counter-based PRNG (splitmix64) used for every synthetic tensor.

The parity contract needs weights / latents / Gumbel draws that
are identical in the build container (where the reference is imported to make
golden fixtures) and on the GPU box (where only the oracle and the HIP path
run).  torch.randn is not stable across machines, so everything comes from
this integer pipeline instead.

* ``uniform``  : exact (integer -> float64 multiply), bit-identical everywhere.
* ``normal``   : Box-Muller in float64 then cast to float32 (libm dependent in
                 the last float64 ulp; used only for *inputs*, never for weights,
                 and small-case inputs are stored inside the fixtures).
* ``exponential``: -log1p(-u) in float64 -> float32 (same remark).
"""
from __future__ import annotations

import zlib

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def key_seed(seed: int, name: str) -> int:
    """Stable 64-bit stream id for (seed, tensor name)."""
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    h2 = zlib.adler32(name.encode("utf-8")) & 0xFFFFFFFF
    return ((int(seed) & 0xFFFFFFFF) * 0x100000001B3 + (h << 32 | h2)) & 0xFFFFFFFFFFFFFFFF


def bits64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n splitmix64 outputs of stream ``seed`` starting at counter ``offset``."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, n: int, lo: float = 0.0, hi: float = 1.0, offset: int = 0) -> np.ndarray:
    """float32 uniform in [lo, hi); exact arithmetic, machine independent."""
    u = (bits64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).astype(np.float32)


def _u01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    return (bits64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float32 N(0,1) by Box-Muller."""
    m = (n + 1) // 2
    u = _u01(seed, 2 * m, 2 * offset)
    u1 = 1.0 - u[0::2]  # (0,1]
    u2 = u[1::2]
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.empty(2 * m, dtype=np.float64)
    out[0::2] = r * np.cos(2.0 * np.pi * u2)
    out[1::2] = r * np.sin(2.0 * np.pi * u2)
    return out[:n].astype(np.float32)


def exponential(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float32 Exp(1) draws, strictly positive."""
    u = _u01(seed, n, offset)
    e = -np.log1p(-u)
    e = np.maximum(e, 1e-30)
    return e.astype(np.float32)


def randint(seed: int, n: int, lo: int, hi: int, offset: int = 0) -> np.ndarray:
    """int64 uniform integers in [lo, hi)."""
    return (lo + (bits64(seed, n, offset) % np.uint64(hi - lo)).astype(np.int64)).astype(np.int64)


# ---------------------------------------------------------------------------
# The library's own router-noise stream (csrc/elementwise.hip: gumbel_draw), restated on the host so that a production run
# (noise drawn on the device, keyed by (seed, global clip, evaluation, branch, block, gate, token, slot)) can be replayed by
# the CPU oracle.  The integer pipeline is exact; u -> Exp(1) uses float32 log1p like the kernel (a last-ulp libm difference
# moves a Gumbel value by ~1e-7 relative: it can flip a hard route only at an exact near-tie).
# ---------------------------------------------------------------------------


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def device_router_exponentials(seed: int, clip: int, nfe: int, branch: int, block: int, gate: int, T: int, width: int) -> np.ndarray:
    """Exp(1) draws E [T, width] float32 with  -log(E) == the Gumbel value gumbel_draw() returns for the same key."""
    u64 = np.uint64
    with np.errstate(over="ignore"):
        key = _splitmix64(u64(seed & 0xFFFFFFFFFFFFFFFF) ^ _splitmix64(u64(clip) * _GOLD + u64(0x1234567)))
        key = _splitmix64(key + (u64((nfe * 2 + branch) << 20)) + (u64(block) << u64(8)) + u64(gate))
        idx = np.arange(1, T * width + 1, dtype=np.uint64)
        z = _splitmix64(key + idx * _GOLD)
    u = ((z >> u64(40)) + u64(1)).astype(np.float32) * np.float32(1.0 / 16777218.0)
    ex = np.maximum(-np.log1p(-u, dtype=np.float32), np.float32(1e-30))
    return ex.reshape(T, width)
