"""Build libversband_hip.so in-tree with hipcc for gfx950 (no GPU needed: cross-compiles).

    python -m versband_amd.build [--force]
    VB_BUILD_EXPERIMENTS=1 python -m versband_amd.build      # + ablation instances and the measured-slower kernels (8-wave GEMM,
                                                             #   fused score/router): tools/ A-B scripts only, never the product

Objects are rebuilt per translation unit (each .o is stamped with the hash of its source, every header and the flags), the library's
source digest is compiled in (vb_source_digest) so the loader can tell a stale binary from a current one without a build directory.
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
SOURCES = ["gemm_bf16.hip", "attention.hip", "conv1d_f32.hip", "conv1d_f32g.hip", "conv1d_f32w.hip", "respair_x3.hip", "respair_f32.hip", "respair_f32w.hip", "t5.hip", "melnet.hip", "elementwise.hip", "rowlin.hip",
           "engine.hip"]
EXPERIMENT_SOURCES = ["score_router.hip"]
EXPERIMENTS = bool(os.environ.get("VB_BUILD_EXPERIMENTS"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + (["-DVB_EXPERIMENTS"] if EXPERIMENTS else [])
# the kernels whose loops read or rescale their MFMA accumulators keep them in VGPRs (hipcc otherwise parks them in AGPRs and copies them
# to VGPRs and back around every ring step / key tile / window chunk: 32-128 v_accvgpr moves per 8-32 MFMAs, and VALU instructions between a
# SIMD's MFMAs cost matrix-pipe time).  VB_BUILD_AGPR=a.hip,b.hip builds the named files without the flag (the A/B of round 4).
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
_AGPR = set(filter(None, os.environ.get("VB_BUILD_AGPR", "").split(",")))
EXTRA_FLAGS = {f: list(_VGPR_FORM) for f in ("conv1d_f32g.hip", "conv1d_f32w.hip", "respair_f32.hip", "respair_f32w.hip", "attention.hip") if f not in _AGPR}
# (measured without effect on the split-bf16 kernels - conv1d_f32.hip, respair_x3.hip: 28.96 / 14.24 ms per pass with the flag, 28.91 / 14.34 without -
#  whose accumulator copies sit in the per-chunk window staging, off the critical path; attention: 12.1 -> 10.9 ms per pass)
MARKER = b"VB_SOURCE_DIGEST="
PUBLIC_HEADER = os.path.join(HERE, "..", "include", "versband_hip.h")


def _sources():
    return SOURCES + (EXPERIMENT_SOURCES if EXPERIMENTS else [])


def sources_present() -> bool:
    return all(os.path.isfile(os.path.join(CSRC, f)) for f in SOURCES) and os.path.isfile(PUBLIC_HEADER)


def source_digest() -> str:
    """sha256 over every file of csrc/ (not csrc/build), the public header and the compile flags"""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(b"../../include/versband_hip.h")
    h.update(open(PUBLIC_HEADER, "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def library_digest(path: str = LIB):
    """the digest compiled into a built library (read from the file, nothing is mapped); None when it carries none"""
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    i = data.find(MARKER)
    if i < 0:
        return None
    return data[i + len(MARKER):i + len(MARKER) + 64].decode("ascii", "replace")


def is_current() -> bool:
    """the in-tree library exists and was built from exactly the sources on disk"""
    return os.path.exists(LIB) and library_digest(LIB) == source_digest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if not force and is_current():
        return LIB
    # one builder at a time: N ranks of a bench / test launch may all find a stale library at once and must not link over each other
    import fcntl
    lock = open(os.path.join(OBJ, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and is_current():      # another process built it while we waited
            return LIB
        return _build_locked(source_digest(), force, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _headers_hash() -> bytes:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(PUBLIC_HEADER, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.digest()


def _build_locked(dg: str, force: bool, verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hh = _headers_hash()

    def cc(src):
        out = os.path.join(OBJ, src.replace(".hip", ".o"))
        stamp = out + ".sha"
        extra = EXTRA_FLAGS.get(src, [])
        want = hashlib.sha256(hh + " ".join(extra).encode() + open(os.path.join(CSRC, src), "rb").read()).hexdigest()
        if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == want:
            return out
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(want)
        return out
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        objs = list(ex.map(cc, srcs))
    # the digest's own translation unit (host code only)
    dsrc = os.path.join(OBJ, "vb_digest.cpp")
    with open(dsrc, "w") as f:
        f.write('extern "C" const char* vb_source_digest(void) { static const char m[] = "%s%s"; return m + %d; }\n'
                % (MARKER.decode(), dg, len(MARKER)))
    dobj = os.path.join(OBJ, "vb_digest.o")
    r = subprocess.run([hipcc, "-O1", "-fPIC", "-x", "c++", "-c", dsrc, "-o", dobj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("digest unit failed:\n" + r.stderr)
    tmp = LIB + f".tmp{os.getpid()}"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs, dobj, "-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    os.replace(tmp, LIB)                     # atomic: a process that already mapped the old library keeps its inode
    if verbose:
        print(f"[versband_amd] built {LIB}" + (" (experiments build)" if EXPERIMENTS else ""))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
