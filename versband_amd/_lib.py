"""ctypes binding of libversband_hip.so (include/versband_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call
fails, an exception is raised.  ``load()`` builds the library with hipcc when
the in-tree .so is absent (hipcc cross-compiles gfx950 without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libversband_hip.so")
VB_MAX_DEPTH = 16

c_void_p, c_int, c_float, c_i64, c_u64, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64, C.c_size_t


class DitConfig(C.Structure):
    _fields_ = [(n, c_int) for n in ("in_channels", "hidden", "heads", "depth", "num_experts", "ffn_hidden", "context_dim",
                                      "ori_dim", "max_len", "np")] + [("norm_eps", c_float)]


BLOCK_FIELDS = ["wqkv", "wo", "wq_m", "wo_m", "w13", "w2", "w13f", "w2f", "wky", "wvy", "wk_m", "wv_m", "wqt_s",
                "bq_m", "bo_m", "bk_m", "bv_m", "bq_s", "attn_norm_w", "ffn_norm_w", "y_norm_w", "cross_w", "wcg", "bcg", "wag", "bag"]


class DitBlockWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in BLOCK_FIELDS]


TOP_FIELDS = ["t_freq_table", "t_mlp0_w", "t_mlp0_b", "t_mlp2_w", "t_mlp2_b", "adaln_w", "adaln_b", "adaln_wp", "hl_w", "hl_b",
              "proj_in_w", "proj_in_b", "proj_in_w3", "final_w", "final_b", "final_wp", "rope_cos", "rope_sin", "midi_emb", "beats_emb",
              "midi_conv_w", "midi_conv_b", "beats_conv_w", "beats_conv_b", "final_proj_w", "final_proj_b", "midi_conv_w3", "beats_conv_w3", "final_proj_w3",
              "c_emb0", "c_emb0_b", "c_emb2", "c_emb2_b", "c_ln_w", "c_ln_b", "cap_ln_w", "cap_ln_b", "cap_lin_w", "cap_lin_b"]


class DitWeights(C.Structure):
    _fields_ = [("blocks", DitBlockWeights * VB_MAX_DEPTH)] + [(n, c_void_p) for n in TOP_FIELDS]


VB_T5_MAX_LAYERS = 48
T5_LAYER_FIELDS = ["ln0", "wqkv", "wo", "ln1", "wi", "wo_ff"]


class T5Config(C.Structure):
    _fields_ = [(n, c_int) for n in ("vocab", "d_model", "d_kv", "heads", "d_ff", "layers")] + [("eps", c_float)]


class T5Layer(C.Structure):
    _fields_ = [(n, c_void_p) for n in T5_LAYER_FIELDS]


class T5Weights(C.Structure):
    _fields_ = [("embed", c_void_p), ("pos_bias", c_void_p), ("pos_len", c_int), ("final_ln", c_void_p), ("ones", c_void_p),
                ("layers", T5Layer * VB_T5_MAX_LAYERS)]


class Noise(C.Structure):
    _fields_ = [("g1", c_void_p), ("g2", c_void_p), ("g3", c_void_p), ("seed", c_u64), ("clip_base", c_i64), ("nfe", c_int)]


class MelConfig(C.Structure):
    _fields_ = [("n_fft", c_int), ("hop", c_int), ("n_mels", c_int)]


class BufDesc(C.Structure):
    _fields_ = [("channels", c_int), ("tmul", c_int), ("square", c_int)]


class NetOp(C.Structure):
    _fields_ = ([("kind", c_int)] + [(n, c_int) for n in ("x", "out", "res", "stats", "w_buf")]
                + [(n, c_void_p) for n in ("w", "bias", "gn_gamma", "gn_beta")]
                + [(n, c_int) for n in ("Ci", "Co", "ksize", "dil", "pad", "upsample2", "in_act", "out_act", "out_transposed",
                                        "tr_stride", "tr_pad", "tr_k", "gn_groups")]
                + [(n, c_float) for n in ("in_slope", "out_slope", "alpha", "beta", "acc_scale")]
                + [("w_x3", c_void_p), ("ci_pad", c_int), ("w2_x3", c_void_p), ("bias2", c_void_p), ("in_stride", c_int), ("in_phase", c_int), ("x_planes", c_int)])


OP_CONV, OP_GN_STATS, OP_SOFTMAX_T, OP_SPLIT_PLANES, OP_RESPAIR, OP_GN_APPLY, OP_AA_ACT, OP_XT_PLANES = 0, 1, 2, 3, 4, 5, 6, 7
ACT_NONE, ACT_LRELU, ACT_GN_SWISH, ACT_TANH, ACT_GN = 0, 1, 2, 3, 4
BUF_INPUT, BUF_OUTPUT = -2, -3
NET_VAE, NET_VOCODER, NET_VAE_ENCODER = 0, 1, 2

# name -> (restype, argtypes); the list doubles as the export check of tests/test_abi.py
P = c_void_p
PROTOTYPES = {
    "vb_ctx_create": (c_int, [c_int, C.POINTER(P)]),
    "vb_ctx_destroy": (c_int, [P]),
    "vb_last_error": (C.c_char_p, []),
    "vb_abi_version": (c_int, []),
    "vb_tune_reload": (None, []),
    "vb_prof_enable": (c_int, [c_int]),
    "vb_prof_read": (c_int, [c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_i64), C.POINTER(c_i64)]),
    "vb_dit_load": (c_int, [P, C.POINTER(DitConfig), C.POINTER(DitWeights)]),
    "vb_dit_cond_bytes": (c_size_t, [C.POINTER(DitConfig), c_int, c_int, c_int, c_int]),
    "vb_dit_workspace_bytes": (c_size_t, [C.POINTER(DitConfig), c_int, c_int, c_int, c_int]),
    "vb_dit_precompute_cond": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "vb_dit_forward": (c_int, [P, P, P, P, C.POINTER(Noise), c_int, c_int, c_int, c_int, P, P, P, P]),
    "vb_source_digest": (C.c_char_p, []),
    "vb_has_experiments": (c_int, []),
    "vb_euler_cfg_step": (c_int, [P, P, c_int, c_i64, c_float, c_float, c_int, P]),
    "vb_sample_cfg": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_float, C.POINTER(Noise), P, P, P]),
    "vb_sample_graphs": (c_int, [P]),
    "vb_net_load": (c_int, [P, c_int, C.POINTER(NetOp), c_int, C.POINTER(BufDesc), c_int, c_int, c_int, c_int, c_int]),
    "vb_net_workspace_bytes": (c_size_t, [P, c_int, c_int, c_int]),
    "vb_vae_decode": (c_int, [P, P, c_int, c_int, P, P, P]),
    "vb_vae_encode": (c_int, [P, P, c_int, c_int, P, P, P]),
    "vb_t5_load": (c_int, [P, C.POINTER(T5Config), C.POINTER(T5Weights)]),
    "vb_t5_workspace_bytes": (C.c_size_t, [C.POINTER(T5Config), c_int, c_int]),
    "vb_t5_encode": (c_int, [P, P, c_int, c_int, P, P, P]),
    "vb_hifigan_forward": (c_int, [P, P, c_int, c_int, P, P, P]),
    "vb_crossfade_windows": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "vb_hifigan_forward_chunked": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    "vb_melnet_load": (c_int, [P, C.POINTER(MelConfig), P, P]),
    "vb_melnet_frames": (c_int, [C.POINTER(MelConfig), c_int, c_int]),
    "vb_melnet_workspace_bytes": (C.c_size_t, [C.POINTER(MelConfig), c_int, c_int, c_int]),
    "vb_melnet_forward": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P]),
    "vb_rmsnorm_modulate": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P, c_int, P]),
    "vb_router_top1": (c_int, [P, P, c_int, c_int, P, P]),
    "vb_route_bucket_scratch_ints": (c_int, [c_int, c_int]),
    "vb_route_bucket": (c_int, [P, P, c_int, c_int, P, P, P]),
    "vb_route_bucket_pairs": (c_int, [P, P, c_int, c_int, P, P, P, P, P]),
    "vb_gemm_bf16": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "vb_grouped_swiglu": (c_int, [P, P, P, c_int, c_int, P, P, P, c_int, c_int, c_int, P, P, P]),
    "vb_attention": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "vb_conv1d_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_float, P, P, P, c_int, P]),
    "vb_conv1d_f32_mf": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P, c_float, c_float, P, P]),
    "vb_respair_f32_mf": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, P, P]),
    "vb_respair_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, P, P]),
    "vb_fill_gumbel": (c_int, [P, c_int, c_int, c_int, c_int, c_u64, c_i64, c_int, c_int, c_int, P]),
    "vb_cast_planes": (c_int, [P, c_i64, P, c_int, P]),
}

_lib = None


class VersbandError(RuntimeError):
    pass


def load(build_if_missing: bool = True):
    """dlopen the in-tree library (build it first if absent)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime: it must be the first libamdhip64 in the process, otherwise this
    # library would bind /opt/rocm's copy and run against a second, device-less runtime instance
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from .build import build, library_digest, source_digest, sources_present
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise VersbandError(f"{LIB_PATH} is missing: run `python -m versband_amd.build`")
        if not _hipcc_available():
            raise VersbandError(f"{LIB_PATH} is missing and hipcc is not available to build it")
        build()
    elif sources_present():
        # never run a binary built from other sources than the ones on disk: an ABI struct mismatch would be memory corruption, not an
        # error.  The digest travels INSIDE the library (vb_source_digest; read from the file here, before anything is mapped), so a
        # prebuilt .so shipped without csrc/build - or without any sources: deployment image, wheel - loads as it is, and a stale
        # one is named as such instead of being rebuilt behind the caller's back when build_if_missing is False.
        have, want = library_digest(LIB_PATH), source_digest()
        if have != want:
            if not build_if_missing:
                raise VersbandError(f"{LIB_PATH} is stale: built from sources {str(have)[:16]}..., the tree holds {want[:16]}... "
                                    "(run `python -m versband_amd.build`)")
            if not _hipcc_available():
                raise VersbandError(f"{LIB_PATH} is stale and hipcc is not available to rebuild it")
            build()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _hipcc_available() -> bool:
    """the compiler build() would run: $HIPCC as a path or as a name on PATH (build() hands it to subprocess either way)"""
    import shutil
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return os.path.exists(hipcc) or shutil.which(hipcc) is not None


def set_tuning(**knobs):
    """Flip VB_* tuning knobs at run time (tools / A-B tests): the library reads its environment once per process, so the
    change is followed by vb_tune_reload().  set_tuning(VB_GEMM_TILE="33"); a value of None removes the variable."""
    for k, v in knobs.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    load().vb_tune_reload()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().vb_last_error()
        raise VersbandError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "non-contiguous tensor handed to the C ABI"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
