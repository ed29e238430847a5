"""The C-ABI library loads and exports every symbol include/versband_hip.h declares (no compute, no GPU)."""
import os
import re

from versband_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "versband_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vb_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    decl = _declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(L.PROTOTYPES) == decl, set(L.PROTOTYPES) ^ set(decl)


def test_library_exports_every_declared_symbol():
    lib = L.load()                      # builds with hipcc if the in-tree .so is absent
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.vb_abi_version() == 2
    assert lib.vb_last_error() is not None


def test_struct_layouts_match_header_field_order():
    src = open(os.path.join(ROOT, "include", "versband_hip.h")).read()
    blk = re.search(r"typedef struct \{(.*?)\} vb_dit_block_weights;", src, flags=re.S).group(1)
    blk = re.sub(r"/\*.*?\*/", "", blk, flags=re.S)
    names = re.findall(r"\*\s*([a-z0-9_]+)\s*;", blk)
    assert names == L.BLOCK_FIELDS
    top = re.search(r"vb_dit_block_weights blocks\[VB_MAX_DEPTH\];(.*?)\} vb_dit_weights;", src, flags=re.S).group(1)
    top = re.sub(r"/\*.*?\*/", "", top, flags=re.S)
    assert re.findall(r"\*\s*([a-z0-9_]+)\s*;", top) == L.TOP_FIELDS
    op = re.search(r"typedef struct \{\s*int kind;(.*?)\} vb_net_op;", src, flags=re.S).group(1)
    op = re.sub(r"/\*.*?\*/", "", op, flags=re.S)
    fields = ["kind"] + [n.strip().lstrip("*") for decl in re.findall(r"(?:int|const float\*|const void\*|float)\s+([^;]+);", op)
                         for n in decl.replace("const float*", "").replace("const void*", "").split(",")]
    assert fields == [f[0] for f in L.NetOp._fields_], fields


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from versband_amd.engine import Context
    with pytest.raises(L.VersbandError):
        Context("cuda:0")


def test_library_carries_the_digest_of_its_sources(monkeypatch):
    """the loader's stale check reads the digest compiled into the .so: current after a build; a library whose digest
    differs from the tree is refused - not silently rebuilt - when build_if_missing is False"""
    import pytest
    from versband_amd import build as B
    lib = L.load()
    assert B.library_digest() == B.source_digest() == lib.vb_source_digest().decode()
    assert lib.vb_has_experiments() in (0, 1)
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(B, "source_digest", lambda: "0" * 64)
    with pytest.raises(L.VersbandError, match="stale"):
        L.load(build_if_missing=False)


def _product_knobs():
    """VB_* environment switches the PRODUCT library reads (engine.hip:tune_load, outside the VB_EXPERIMENTS block)"""
    src = open(os.path.join(ROOT, "versband_amd", "csrc", "engine.hip")).read()
    body = src[src.index("static void tune_load()"):]
    body = body[:body.index("g_tune = t;")]
    product = re.sub(r"#ifdef VB_EXPERIMENTS.*?#endif", "", body, flags=re.S)
    return sorted(set(re.findall(r'"(VB_[A-Z0-9_]+)"', product))), sorted(set(re.findall(r'"(VB_[A-Z0-9_]+)"', body)))


def test_every_product_knob_is_documented_and_flipped_by_a_test():
    """A switch the product library reads is either named in INTEGRATION.md (what an integrator may meet) and exercised by a test or a
    tool, or it does not belong in the product build (round 2 shipped 22 of them, half undocumented)."""
    product, _ = _product_knobs()
    assert len(product) >= 10
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read() + open(os.path.join(ROOT, "DESIGN.md")).read()
    used = ""
    for d in ("tests", "tools"):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".py", ".sh")):
                used += open(os.path.join(ROOT, d, f)).read()
    used += open(os.path.join(ROOT, "bench.py")).read()
    for k in product:
        assert k in doc, f"{k} is read by the product library but documented nowhere"
        assert k in used, f"{k} is read by the product library but no test or tool sets it"


def test_no_kernel_of_the_product_library_uses_scratch(tmp_path):
    """Round-3 verdict, weak item 9: instances that spill (576-640 B of scratch at 254-256 VGPRs) shipped in the product library.
    Every kernel of the built libversband_hip.so must run out of registers alone: private_segment_fixed_size = 0 and no spilled VGPR
    in the code objects' metadata (read from the file, nothing is run; SGPRs parked in VGPR lanes never touch memory and are allowed)."""
    import shutil
    import subprocess
    import pytest
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.isfile(os.path.join(llvm, "llvm-objdump")) and os.path.isfile(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("ROCm LLVM tools not installed")
    if not os.path.isfile(L.LIB_PATH):
        pytest.skip("library not built")
    so = shutil.copy(L.LIB_PATH, str(tmp_path))
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", so], cwd=str(tmp_path), check=True, capture_output=True)
    objs = [f for f in os.listdir(str(tmp_path)) if "hipv4-amdgcn" in f]
    assert objs, "no device code object found in the library"
    kernels, bad = 0, []
    for f in objs:
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", os.path.join(str(tmp_path), f)], check=True, capture_output=True,
                               text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            vals = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("private_segment_fixed_size", "vgpr_spill_count")}
            kernels += 1
            if any(vals.values()):
                bad.append((name.group(1) if name else "?", vals))
    assert kernels > 100, f"only {kernels} kernels found - metadata layout changed?"
    assert not bad, f"kernels with scratch / spills: {bad}"
