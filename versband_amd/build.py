"""Build libversband_hip.so in-tree with hipcc for gfx950 (no GPU needed: cross-compiles).

    python -m versband_amd.build [--force]
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libversband_hip.so")
SOURCES = ["gemm_bf16.hip", "attention.hip", "conv1d_f32.hip", "respair_x3.hip", "t5.hip", "melnet.hip", "elementwise.hip", "score_router.hip", "rowlin.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _digest() -> str:
    h = hashlib.sha256()
    files = sorted(os.listdir(CSRC)) + ["../../include/versband_hip.h"]
    for f in files:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_current() -> bool:
    """the in-tree library exists and was built from exactly the sources on disk"""
    stamp = os.path.join(OBJ, "digest.txt")
    return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == _digest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "digest.txt")
    dg = _digest()
    if not force and is_current():
        return LIB
    # one builder at a time: N ranks of a bench / test launch may all find a stale stamp at once and must not link over each other
    import fcntl
    lock = open(os.path.join(OBJ, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and is_current():      # another process built it while we waited
            return LIB
        return _build_locked(dg, stamp, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(dg: str, stamp: str, verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def cc(src):
        out = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return out
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    tmp = LIB + f".tmp{os.getpid()}"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    os.replace(tmp, LIB)                     # atomic: a process that already mapped the old library keeps its inode
    with open(stamp, "w") as f:
        f.write(dg)
    if verbose:
        print(f"[versband_amd] built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
