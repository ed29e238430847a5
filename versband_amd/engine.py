"""Python host side over the C ABI: device context, DiT engine, conv-net programs.

torch is used for device memory, streams and checkpoint tensors only; every
compute step of the hot path is a call into libversband_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import pack
from .synth import DiTConfig, HifiGanConfig, VAEConfig

Tensor = torch.Tensor


def _require_gpu(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise L.VersbandError("versband_amd runs on an MI355X only: no HIP device is visible (there is no CPU fallback)")
    return device


class Context:
    """One vb_ctx per (process, device) - mirrors the reference's one process per GPU."""

    def __init__(self, device="cuda:0"):
        self.device = _require_gpu(device)
        self.lib = L.load()
        self.handle = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        torch.cuda.set_device(idx)
        L.check(self.lib.vb_ctx_create(idx, C.byref(self.handle)), "vb_ctx_create")
        self._keep: List[object] = []

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.vb_ctx_destroy(self.handle)
        except Exception:
            pass


# ---------------------------------------------------------------------------
# DiT
# ---------------------------------------------------------------------------


class DiTEngine:
    """TxtFlagLargeImprovedDiTV2 (vocal2music_moe.py:477-520) + CFMSampler hot loop on the GPU.

    precision "bf16"  : plain bf16 MFMA operands (production / benchmark)
    precision "split" : bf16x3 split operands (fp32-class, parity tests)"""

    def __init__(self, ctx: Context, cfg: DiTConfig, state_dict: Dict[str, Tensor], precision: str = "bf16", share: "DiTEngine" = None):
        assert precision in ("bf16", "split")
        ctx = Context(ctx.device)        # a vb_ctx holds ONE loaded DiT: every engine owns its handle
        self.ctx, self.cfg, self.precision = ctx, cfg, precision
        self.np = 2 if precision == "split" else 1
        dev = ctx.device
        # engines of one process that serve different sub-batches share ONE packed copy of the weights (`share`)
        # (`share` must be an engine of the SAME configuration, precision and checkpoint: anything else would hand vb_dit_load another
        #  model's pointer layout silently)
        if share is not None:
            if share.cfg != cfg or share.precision != precision:
                raise ValueError(f"DiTEngine(share=...): the shared engine was built for {share.cfg} / {share.precision}, this one asks for {cfg} / {precision}")
            if share._sd_id is not state_dict and share._sd_id is not None and state_dict is not None:
                same = set(share._sd_id.keys()) == set(state_dict.keys()) and all(
                    share._sd_id[k].data_ptr() == state_dict[k].data_ptr() for k in list(state_dict.keys())[:8])
                if not same:
                    raise ValueError("DiTEngine(share=...): the shared engine holds another checkpoint")
        self._sd_id = state_dict
        self.packed = share.packed if share is not None else pack.pack_dit(state_dict, cfg, self.np, dev)
        self.ccfg = L.DitConfig(cfg.in_channels, cfg.hidden_size, cfg.num_heads, cfg.depth, cfg.num_experts, cfg.ffn_hidden,
                                cfg.context_dim, cfg.ori_dim, cfg.max_len, self.np, cfg.norm_eps)
        w = L.DitWeights()
        for name in L.TOP_FIELDS:
            t = self.packed["top"][name]
            setattr(w, name, t.data_ptr() if t is not None else None)
        for i, b in enumerate(self.packed["blocks"]):
            for name in L.BLOCK_FIELDS:
                setattr(w.blocks[i], name, b[name].data_ptr())
        self.cw = w
        L.check(ctx.lib.vb_dit_load(ctx.handle, C.byref(self.ccfg), C.byref(w)), "vb_dit_load")
        self._ws: Optional[Tensor] = None
        self._ws_key = None
        self._checked: Dict[tuple, tuple] = {}                   # range-checked resident index tracks: key -> (midi, beats, versions), references held
        self._tables: Dict[tuple, Tuple[Tensor, Tensor]] = {}    # device copies of the (t_idx, dt) step tables
        self._pcond: Dict[tuple, list] = {}                      # persistent conditioning buffers: key -> [buffer, generation]
        self._xbuf: Dict[tuple, Tensor] = {}                     # engine-owned sampler state (stable address for graph replay)

    # -- buffers -----------------------------------------------------------
    def _workspace(self, B, nb, T, Lc) -> Tensor:
        key = (B, nb, T, Lc)
        if self._ws_key != key:
            n = self.ctx.lib.vb_dit_workspace_bytes(C.byref(self.ccfg), B, nb, T, Lc)
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.ctx.device)
            self._ws_key = key
        return self._ws

    # -- API ---------------------------------------------------------------
    def precompute_cond(self, t5: Tensor, midi: Tensor, beats: Tensor, T: int, persistent: bool = False) -> dict:
        """t5 [nb*B, L, ori] (cond rows then uncond rows), midi/beats [B,1,T_mel] or [B,T_mel] int64.
        persistent=True writes into an engine-owned buffer that the NEXT persistent precompute of the same shape overwrites (a
        serving loop: precompute -> sample -> discard): stable device addresses are what lets vb_sample_cfg replay its captured
        hipGraph.  A stale handle is refused loudly (generation check) instead of sampling from overwritten conditioning."""
        dev = self.ctx.device
        t5 = t5.to(dev, torch.float32).contiguous()
        midi_in, beats_in = midi, beats
        midi = midi.to(dev, torch.int64).reshape(midi.shape[0], -1).contiguous()
        beats = beats.to(dev, torch.int64).reshape(beats.shape[0], -1).contiguous()
        B, T_mel = midi.shape
        Beff, Lc, _ = t5.shape
        nb = Beff // B
        assert nb * B == Beff and nb in (1, 2)
        # range check of the embedding indices (embed_t_kernel reads table[idx * D ..] unclamped): ONE device->host read per new pair
        # of tracks.  The check is skipped only when the CALLER's own tensors are what the kernel will read (resident int64 contiguous
        # device tensors: .to() / reshape / contiguous returned views of the same storage) AND were checked before at the same
        # version; the cache entry holds a reference to them, so their addresses cannot be recycled for other data while the entry
        # lives.  Converted temporaries (CPU, int32, strided inputs) are checked on every call.
        resident = (midi.data_ptr() == midi_in.data_ptr() and beats.data_ptr() == beats_in.data_ptr() and
                    midi_in.dtype == torch.int64 and beats_in.dtype == torch.int64 and midi_in.device == midi.device and
                    beats_in.device == beats.device and midi_in.is_contiguous() and beats_in.is_contiguous())
        key = (midi.data_ptr(), beats.data_ptr(), midi.numel(), beats.numel())
        hit = self._checked.get(key) if resident else None
        # (a view of the same storage shares its version counter, and the held reference keeps that storage - hence the address - alive)
        if hit is None or hit[2] != (midi_in._version, beats_in._version):
            lim = torch.stack([midi.min(), midi.max(), beats.min(), beats.max()]).tolist()
            if lim[0] < 0 or lim[1] >= 130 or lim[2] < 0 or lim[3] >= 3:
                raise IndexError("midi/beats index out of range of the embedding tables (130 / 3 rows)")
            if resident:
                if len(self._checked) > 64:
                    self._checked.clear()
                self._checked[key] = (midi_in, beats_in, (midi_in._version, beats_in._version))
        n = self.ctx.lib.vb_dit_cond_bytes(C.byref(self.ccfg), B, nb, T, Lc)
        gen = None
        if persistent:
            key = (B, nb, T, Lc)
            slot = self._pcond.get(key)
            if slot is None or slot[0].numel() != n:
                slot = [torch.empty(n, dtype=torch.uint8, device=dev), 0]
                self._pcond[key] = slot
            slot[1] += 1
            cond, gen = slot[0], slot[1]
        else:
            cond = torch.empty(n, dtype=torch.uint8, device=dev)
        ws = self._workspace(B, nb, T, Lc)
        L.check(self.ctx.lib.vb_dit_precompute_cond(self.ctx.handle, L.ptr(t5), L.ptr(midi), L.ptr(beats), B, nb, T, T_mel, Lc,
                                                    L.ptr(cond), L.ptr(ws), L.stream_ptr()), "vb_dit_precompute_cond")
        return {"buf": cond, "B": B, "nb": nb, "T": T, "L": Lc, "gen": gen}

    def _check_cond(self, cond: dict):
        if cond.get("gen") is not None:
            slot = self._pcond.get((cond["B"], cond["nb"], cond["T"], cond["L"]))
            if slot is None or slot[1] != cond["gen"] or slot[0] is not cond["buf"]:
                raise L.VersbandError("stale persistent conditioning handle: a later precompute_cond(persistent=True) of the same shape "
                                      "overwrote this buffer")

    def _noise_struct(self, noise, seed, clip_base, nfe):
        ns = L.Noise()
        keep = None
        if noise is not None:
            g1, g2, g3 = [t.to(self.ctx.device, torch.float32).contiguous() for t in noise]
            ns.g1, ns.g2, ns.g3 = g1.data_ptr(), g2.data_ptr(), g3.data_ptr()
            keep = (g1, g2, g3)
        ns.seed, ns.clip_base, ns.nfe = seed, clip_base, nfe
        return ns, keep

    def forward(self, x: Tensor, t_idx: Tensor, cond: dict, noise=None, seed: int = 0, clip_base: int = 0, nfe: int = 0,
                return_routes: bool = False):
        """x [B,C,T] f32, t_idx int64 [nb*B]; noise = (g1 [depth,rows,2], g2 [depth,rows,E], g3 [depth,rows,E]) Gumbel
        draws or None -> v [nb*B, C, T] (+ routes int32 [depth,2,rows])."""
        dev = self.ctx.device
        B, nb, T, Lc = cond["B"], cond["nb"], cond["T"], cond["L"]
        self._check_cond(cond)
        x = x.to(dev, torch.float32).contiguous()
        t_idx = t_idx.to(dev, torch.int64).contiguous()
        assert x.shape == (B, self.cfg.in_channels, T) and t_idx.numel() == nb * B
        v = torch.empty(nb * B, self.cfg.in_channels, T, dtype=torch.float32, device=dev)
        routes = torch.empty(self.cfg.depth, 2, nb * B * T, dtype=torch.int32, device=dev) if return_routes else None
        ns, keep = self._noise_struct(noise, seed, clip_base, nfe)
        ws = self._workspace(B, nb, T, Lc)
        L.check(self.ctx.lib.vb_dit_forward(self.ctx.handle, L.ptr(x), L.ptr(t_idx), L.ptr(cond["buf"]), C.byref(ns), B, nb, T, Lc,
                                            L.ptr(v), L.ptr(routes), L.ptr(ws), L.stream_ptr()), "vb_dit_forward")
        return (v, routes) if return_routes else v

    def sample_cfg(self, x0: Tensor, cond: dict, t_idx_table: Sequence[int], dt_table: Sequence[float], scale: float,
                   noise=None, seed: int = 0, clip_base: int = 0, return_traj: bool = False):
        """n Euler steps with classifier-free guidance; x0 [B,C,T] is not modified."""
        dev = self.ctx.device
        B, nb, T, Lc = cond["B"], cond["nb"], cond["T"], cond["L"]
        self._check_cond(cond)
        xkey = (B, self.cfg.in_channels, T)
        assert tuple(x0.shape) == xkey, (tuple(x0.shape), xkey)
        if xkey not in self._xbuf:
            self._xbuf[xkey] = torch.empty(xkey, dtype=torch.float32, device=dev)
        x = self._xbuf[xkey]            # the state the solver integrates in place; the caller gets a copy
        x.copy_(x0.to(dev, torch.float32))
        n = len(t_idx_table)
        tkey = (tuple(int(v) for v in t_idx_table), tuple(float(v) for v in dt_table))
        if tkey not in self._tables:            # step tables live on the device: nothing on the host has to outlive the async launch
            if len(self._tables) > 16:
                self._tables.clear()
            self._tables[tkey] = (torch.tensor(tkey[0], dtype=torch.int64, device=dev), torch.tensor(tkey[1], dtype=torch.float32, device=dev))
        tt, dd = self._tables[tkey]
        traj = torch.empty(n + 1, *x.shape, dtype=torch.float32, device=dev) if return_traj else None
        ns, keep = self._noise_struct(noise, seed, clip_base, 0)
        ws = self._workspace(B, nb, T, Lc)
        L.check(self.ctx.lib.vb_sample_cfg(self.ctx.handle, L.ptr(x), L.ptr(cond["buf"]), B, nb, T, Lc, n, L.ptr(tt), L.ptr(dd), float(scale),
                                           C.byref(ns), L.ptr(traj), L.ptr(ws), L.stream_ptr()), "vb_sample_cfg")
        x = x.clone()
        return (x, traj) if return_traj else x

    def graphs(self) -> int:
        """instantiated hipGraphs of the sampler loop this engine's context holds (0 = every call ran eagerly)"""
        return int(self.ctx.lib.vb_sample_graphs(self.ctx.handle))


class T5Engine:
    """transformers.T5EncoderModel(input_ids).last_hidden_state on the HIP library: token ids [B,L] ->
    [B,L,d_model] fp32.  `sd` uses the HF state_dict keys; heads / eps / bucket parameters come from the HF config."""

    def __init__(self, ctx: Context, sd: Dict[str, Tensor], num_heads: int = 16, d_kv: int = 64, eps: float = 1e-6, max_len: int = 128,
                 num_buckets: int = 32, max_distance: int = 128):
        ctx = Context(ctx.device)
        self.ctx = ctx
        top, layers = pack.pack_t5(sd, ctx.device, num_heads, max_len, num_buckets, max_distance)
        self._keep = (top, layers)
        d_ff = sd["encoder.block.0.layer.1.DenseReluDense.wi_0.weight"].shape[0]
        self.cfg = L.T5Config(vocab=top["embed"].shape[0], d_model=top["embed"].shape[1], d_kv=d_kv, heads=num_heads, d_ff=d_ff,
                              layers=len(layers), eps=eps)
        w = L.T5Weights()
        w.embed, w.pos_bias, w.pos_len = top["embed"].data_ptr(), top["pos_bias"].data_ptr(), max_len
        w.final_ln, w.ones = top["final_ln"].data_ptr(), top["ones"].data_ptr()
        for i, lw in enumerate(layers):
            for f in L.T5_LAYER_FIELDS:
                setattr(w.layers[i], f, lw[f].data_ptr())
        L.check(ctx.lib.vb_t5_load(ctx.handle, C.byref(self.cfg), C.byref(w)), "vb_t5_load")
        self._ws = None

    def encode(self, ids: Tensor) -> Tensor:
        dev = self.ctx.device
        ids = ids.to(dev, torch.int64).contiguous()
        B, Lc = ids.shape
        n = self.ctx.lib.vb_t5_workspace_bytes(C.byref(self.cfg), B, Lc)
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.uint8, device=dev)
        out = torch.empty(B, Lc, self.cfg.d_model, dtype=torch.float32, device=dev)
        L.check(self.ctx.lib.vb_t5_encode(self.ctx.handle, L.ptr(ids), B, Lc, L.ptr(out), L.ptr(self._ws), L.stream_ptr()), "vb_t5_encode")
        return out


# ---------------------------------------------------------------------------
# conv nets
# ---------------------------------------------------------------------------


class NetBuilder:
    """Flattens a conv network into the vb_net_op list executed by the C++ runtime."""

    def __init__(self, device, precision: str = "split"):
        assert precision in ("split", "fp32", "fp32mf")
        self.device = device
        # "split": bf16x3 MFMA conv kernel (fp32-class); "fp32": exact f32 MFMA kernel; "fp32mf": the fp32 op list with F(2,3) minimal
        # filtering on the stride-1 k = 3 / 5 / 7 / 11 convolutions of >= 128 output channels (one tile shape: 128 co; narrower layers are faster direct) (fp32 products, ~1.45x fewer; conv1d_f32w.hip)
        self.mf = precision == "fp32mf"
        self.precision = "fp32" if self.mf else precision
        self.ops: List[L.NetOp] = []
        self.bufs: List[Tuple[int, int, int]] = []
        self.free: Dict[Tuple[int, int, int], List[int]] = {}
        self.keep: List[Tensor] = []

    def buf(self, channels: int, tmul: int, square: bool = False) -> int:
        key = (channels, tmul, int(square))
        if self.free.get(key):
            return self.free[key].pop()
        self.bufs.append(key)
        return len(self.bufs) - 1

    def release(self, b: int):
        if b >= 0:
            self.free.setdefault(self.bufs[b], []).append(b)

    def _t(self, t: Optional[Tensor]):
        if t is None:
            return None
        t = t.to(self.device, torch.float32).contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def gn_stats(self, x: int, channels: int, groups: int = 32) -> int:
        st = self.buf(2 * groups, 1)
        op = L.NetOp(kind=L.OP_GN_STATS, x=x, out=-1, res=-1, stats=st, w_buf=-1, Ci=channels, gn_groups=groups)
        self.ops.append(op)
        return st

    def split_planes(self, x: int, out: int, rows: int, cols: int):
        """f32 [B][rows][cols] buffer -> split-bf16 planes [2][B][rows][roundup(cols,32)] (-1 = the buffer's time length)."""
        self.ops.append(L.NetOp(kind=L.OP_SPLIT_PLANES, x=x, out=out, res=-1, stats=-1, w_buf=-1, Ci=cols, Co=rows))

    def respair(self, x: int, out: int, ch: int, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, k: int, dil: int, slope: float,
                alpha: float, beta: float):
        """Fused HiFi-GAN ResBlock1 pair (w1/w2 fp32 packed [k][Ci][Co]); narrow stages only (ch = 32 or 64; fp32 mode: 32 / 64 / 128)."""
        if self.precision == "fp32" and self.mf and ch in (32, 64) and k in (3, 7, 11) and not os.environ.get("VB_MF_PAIRS_OFF"):
            # fp32mf: both convolutions of the pair by F(2,3) minimal filtering, intermediate in LDS (respair_f32w.hip); ci_pad = -2 marks the
            # weights as pseudo-taps [P][C][C]
            self.ops.append(L.NetOp(kind=L.OP_RESPAIR, x=x, out=out, res=-1, stats=-1, w_buf=-1, w=self._t(pack.pack_conv_mf(w1.permute(2, 1, 0))),
                                    bias=self._t(b1), bias2=self._t(b2), Ci=ch, Co=ch, ksize=k, dil=dil, in_slope=slope, alpha=alpha, beta=beta,
                                    w_x3=None, w2_x3=self._t(pack.pack_conv_mf(w2.permute(2, 1, 0))), ci_pad=-2))
            return
        if self.precision == "fp32":
            # exact-fp32 pair kernel (respair_f32.hip): the packed fp32 weights go in as they are
            self.ops.append(L.NetOp(kind=L.OP_RESPAIR, x=x, out=out, res=-1, stats=-1, w_buf=-1, w=self._t(w1), bias=self._t(b1),
                                    bias2=self._t(b2), Ci=ch, Co=ch, ksize=k, dil=dil, in_slope=slope, alpha=alpha, beta=beta,
                                    w_x3=None, w2_x3=self._t(w2), ci_pad=ch))
            return
        p1, c1 = pack.pack_conv_x3(w1.to(self.device))
        p2, c2 = pack.pack_conv_x3(w2.to(self.device))
        assert c1 == ch and c2 == ch
        self.keep += [p1, p2]
        self.ops.append(L.NetOp(kind=L.OP_RESPAIR, x=x, out=out, res=-1, stats=-1, w_buf=-1, bias=self._t(b1), bias2=self._t(b2),
                                Ci=ch, Co=ch, ksize=k, dil=dil, in_slope=slope, alpha=alpha, beta=beta, w_x3=p1.data_ptr(),
                                w2_x3=p2.data_ptr(), ci_pad=ch))

    def aa_act(self, x: int, out: int, ch: int, alpha: Tensor, beta: Optional[Tensor], logscale: bool):
        """BigVGAN Activation1d(Snake / SnakeBeta): out = down2(act(up2(x))) (alias_free_torch/act.py)."""
        if getattr(self, "_aa_filter", None) is None:
            self._aa_filter = pack.kaiser_sinc_filter1d(0.25, 0.3, 12)
        a = alpha.float()
        b = a if beta is None else beta.float()
        if logscale:
            a, b = torch.exp(a), torch.exp(b)
        self.ops.append(L.NetOp(kind=L.OP_AA_ACT, x=x, out=out, res=-1, stats=-1, w_buf=-1, Ci=ch, gn_gamma=self._t(a),
                                gn_beta=self._t(1.0 / (b + 1e-9)), w=self._t(self._aa_filter)))

    def softmax_t(self, x: int, out: int):
        self.ops.append(L.NetOp(kind=L.OP_SOFTMAX_T, x=x, out=out, res=-1, stats=-1, w_buf=-1))

    def conv(self, x: int, out: int, Ci: int, Co: int, w: Optional[Tensor] = None, bias: Optional[Tensor] = None, k: int = 1,
             dil: int = 1, pad: int = 0, res: int = -1, stats: int = -1, gamma=None, beta_gn=None, in_act=L.ACT_NONE,
             in_slope=0.0, out_act=L.ACT_NONE, out_slope=0.0, upsample2=0, out_transposed=0, w_buf=-1, alpha=1.0, beta=0.0,
             acc_scale=1.0, tr_stride=1, tr_pad=0, tr_k=0, groups=32, w_buf_planes=False, in_stride=1, in_phase=0, x_is_planes=False):
        w_x3, ci_pad = None, (-1 if w_buf_planes else 0)
        tmp = -1
        x_planes = int(bool(x_is_planes))
        if (self.precision == "split" and not x_planes and w is not None and w_buf == -1 and x >= 0 and Co >= self.XT_MIN_CO and Ci % 32 == 0
                and tr_stride == 1 and in_stride == 1 and pad <= 64 and (k - 1) * dil <= 64 and (k > 1 or in_act != L.ACT_NONE)):
            # wide layer (>= 3 output-channel tiles): the input is activated, split and transposed ONCE (VB_OP_XT_PLANES) and every
            # tile of the convolution DMAs its window from there, instead of each tile redoing norm / swish / split while staging
            tmp = self.xt_planes(x, Ci, stats, gamma, beta_gn, in_act, in_slope, upsample2, groups)
            x, stats, gamma, beta_gn, in_act, x_planes = tmp, -1, None, None, L.ACT_NONE, 1
        if (self.precision == "fp32" and in_act in (L.ACT_GN_SWISH, L.ACT_GN) and stats >= 0 and w is not None and w_buf == -1 and x >= 0
                and Ci % 16 == 0 and Co % 4 == 0 and in_stride == 1 and tr_stride == 1 and not os.environ.get("VB_FP32_NO_PREPASS")):
            # exact-fp32 mode: GroupNorm (+ swish) is applied ONCE by gn_apply_kernel and the DMA-fed fp32 kernel (conv1d_f32g_kernel)
            # reads the activated tensor; the register-staged kernel redid norm + swish + expf for every output-channel tile
            # (12 times on the 1536-channel layers: 2.3 ms for an 85-GFLOP layer).  VB_FP32_NO_PREPASS=1 keeps the old op list (A/B).
            tmp = self.gn_apply(x, Ci, stats, gamma, beta_gn, in_act, groups)
            x, stats, gamma, beta_gn, in_act = tmp, -1, None, None, L.ACT_NONE
        if self.precision == "split" and w is not None and w_buf == -1:
            planes, ci_pad = pack.pack_conv_x3(w.to(self.device))
            self.keep.append(planes)
            w_x3 = planes.data_ptr()
        w_mf = None
        if (self.mf and w is not None and w_buf == -1 and tr_stride == 1 and in_stride == 1 and not upsample2 and not out_transposed and
                k in (3, 5, 7, 11) and (k - 1) * dil <= 60 and dil <= 8 and Ci % 16 == 0 and Co % 4 == 0 and Co >= 64 and
                in_act in (L.ACT_NONE, L.ACT_LRELU) and w.shape == (k, Ci, Co)):
            w_mf = self._t(pack.pack_conv_mf(w.permute(2, 1, 0)))          # (w arrives packed [k][Ci][Co])
        op = L.NetOp(kind=L.OP_CONV, x=x, out=out, res=res, stats=stats, w_buf=w_buf, w=self._t(w), bias=self._t(bias), w2_x3=w_mf,
                     gn_gamma=self._t(gamma), gn_beta=self._t(beta_gn), Ci=Ci, Co=Co, ksize=k, dil=dil, pad=pad,
                     upsample2=upsample2, in_act=in_act, out_act=out_act, out_transposed=out_transposed, tr_stride=tr_stride,
                     tr_pad=tr_pad, tr_k=tr_k, gn_groups=groups, in_slope=in_slope, out_slope=out_slope, alpha=alpha, beta=beta,
                     acc_scale=acc_scale, w_x3=w_x3, ci_pad=ci_pad, in_stride=in_stride, in_phase=in_phase, x_planes=x_planes)
        self.ops.append(op)
        self.release(tmp)

    GN_PREPASS_MIN_CO = 512
    XT_MIN_CO = int(os.environ.get("VB_XT_MIN_CO", "384"))      # (tuning knob; 256 / 128 would pull the wide vocoder stages in)

    def xt_planes(self, x: int, channels: int, stats: int, gamma, beta_gn, act: int, slope: float, upsample2: int, groups: int = 32) -> int:
        """x -> new buffer of pre-activated, split-bf16, transposed planes with zero halo rows (VB_OP_XT_PLANES)."""
        out = self.buf(channels, self.bufs[x][1] * (2 if upsample2 else 1), 3)
        self.ops.append(L.NetOp(kind=L.OP_XT_PLANES, x=x, out=out, res=-1, stats=stats, w_buf=-1, gn_gamma=self._t(gamma),
                                gn_beta=self._t(beta_gn), Ci=channels, gn_groups=groups, in_act=act, in_slope=slope, upsample2=upsample2))
        return out

    def gn_apply(self, x: int, channels: int, stats: int, gamma, beta_gn, act: int, groups: int = 32) -> int:
        """x -> new buffer holding GroupNorm(x) (act = ACT_GN) or swish(GroupNorm(x)) (ACT_GN_SWISH)."""
        out = self.buf(channels, self.bufs[x][1])
        self.ops.append(L.NetOp(kind=L.OP_GN_APPLY, x=x, out=out, res=-1, stats=stats, w_buf=-1, gn_gamma=self._t(gamma),
                                gn_beta=self._t(beta_gn), Ci=channels, gn_groups=groups, in_act=act))
        return out


class ConvNet:
    def __init__(self, ctx: Context, which: int, nb: NetBuilder, in_ch: int, out_ch: int, out_tmul: int, in_tmul: int = 1):
        ctx = Context(ctx.device)        # one program per (vb_ctx, slot): every net owns its handle
        self.ctx, self.which, self.nb = ctx, which, nb
        self.in_ch, self.out_ch, self.out_tmul, self.in_tmul = in_ch, out_ch, out_tmul, in_tmul
        ops = (L.NetOp * len(nb.ops))(*nb.ops)
        bufs = (L.BufDesc * max(1, len(nb.bufs)))(*[L.BufDesc(*b) for b in nb.bufs])
        L.check(ctx.lib.vb_net_load(ctx.handle, which, ops, len(nb.ops), bufs, len(nb.bufs), in_ch, out_ch, in_tmul, out_tmul),
                "vb_net_load")
        self._ws = None
        self._ws_key = None

    def _workspace(self, B, T):
        if self._ws_key != (B, T):
            n = self.ctx.lib.vb_net_workspace_bytes(self.ctx.handle, self.which, B, T)
            self._ws = torch.empty(max(n, 256), dtype=torch.uint8, device=self.ctx.device)
            self._ws_key = (B, T)
        return self._ws

    def run(self, x: Tensor) -> Tensor:
        x = x.to(self.ctx.device, torch.float32).contiguous()
        B, Cin, Tin = x.shape
        assert Cin == self.in_ch, (Cin, self.in_ch)
        if Tin % self.in_tmul:
            raise ValueError(f"input length {Tin} is not a multiple of {self.in_tmul}")
        T = Tin // self.in_tmul
        out = torch.empty(B, self.out_ch, T * self.out_tmul, dtype=torch.float32, device=self.ctx.device)
        ws = self._workspace(B, T)
        fn = {L.NET_VAE: self.ctx.lib.vb_vae_decode, L.NET_VOCODER: self.ctx.lib.vb_hifigan_forward,
              L.NET_VAE_ENCODER: self.ctx.lib.vb_vae_encode}[self.which]
        L.check(fn(self.ctx.handle, L.ptr(x), B, T, L.ptr(out), L.ptr(ws), L.stream_ptr()), "conv net run")
        return out


def _convnet_run_chunked(self, x: Tensor, chunk: int, halo: int) -> Tensor:
    """vocoder only: mel [B,80,T] -> wav [B,1,T*hop] in chunks of `chunk` frames with `halo` frames of context on both sides, the loop,
    the slicing and the stitching inside the library (vb_hifigan_forward_chunked); identical to run() when halo >= the receptive field."""
    assert self.which == L.NET_VOCODER
    x = x.to(self.ctx.device, torch.float32).contiguous()
    B, Cin, T = x.shape
    assert Cin == self.in_ch
    if T <= chunk + 2 * halo:
        return self.run(x)
    tc = chunk + 2 * halo
    out = torch.empty(B, self.out_ch, T * self.out_tmul, dtype=torch.float32, device=self.ctx.device)
    ws = self._workspace(B, tc)
    key = (B, tc)
    if getattr(self, "_chunk_key", None) != key:
        self._chunk_in = torch.empty(B * self.in_ch * tc, dtype=torch.float32, device=self.ctx.device)
        self._chunk_out = torch.empty(B * self.out_ch * tc * self.out_tmul, dtype=torch.float32, device=self.ctx.device)
        self._chunk_key = key
    L.check(self.ctx.lib.vb_hifigan_forward_chunked(self.ctx.handle, L.ptr(x), B, T, chunk, halo, L.ptr(out), L.ptr(ws), L.ptr(self._chunk_in),
                                                    L.ptr(self._chunk_out), L.stream_ptr()), "vb_hifigan_forward_chunked")
    return out


ConvNet.run_chunked = _convnet_run_chunked


def _vae_block_builders(nb: "NetBuilder", g: Dict[str, Tensor]):
    """(cw, resblock, attnblock) emitting ResnetBlock1D (autoencoder1d.py:172-231) / AttnBlock1D (:233-274) ops into nb."""

    def cw(name):
        return pack.pack_conv(g[name + ".weight"]), g[name + ".bias"]

    def resblock(x, cin, tm, p):
        cout = g[p + "conv1.weight"].shape[0]
        st1 = nb.gn_stats(x, cin)
        t1 = nb.buf(cout, tm)
        w, b = cw(p + "conv1")
        kk = g[p + "conv1.weight"].shape[2]      # 3 in the decoder, ddconfig.kernel_size in the encoder (autoencoder1d.py:346-351)
        nb.conv(x, t1, cin, cout, w, b, k=kk, pad=kk // 2, stats=st1, gamma=g[p + "norm1.weight"], beta_gn=g[p + "norm1.bias"],
                in_act=L.ACT_GN_SWISH)
        nb.release(st1)
        st2 = nb.gn_stats(t1, cout)
        sc = x
        if (p + "nin_shortcut.weight") in g:
            sc = nb.buf(cout, tm)
            w, b = cw(p + "nin_shortcut")
            nb.conv(x, sc, cin, cout, w, b)
        out = nb.buf(cout, tm)
        w, b = cw(p + "conv2")
        kk = g[p + "conv2.weight"].shape[2]
        nb.conv(t1, out, cout, cout, w, b, k=kk, pad=kk // 2, res=sc, stats=st2, gamma=g[p + "norm2.weight"], beta_gn=g[p + "norm2.bias"],
                in_act=L.ACT_GN_SWISH)
        nb.release(st2); nb.release(t1)
        if sc != x:
            nb.release(sc)
        nb.release(x)
        return out, cout

    def attnblock(x, c, tm, p):
        # single-head attention over time (AttnBlock1D, autoencoder1d.py): both matrix products run as k=1 convs whose
        # "weights" are per-batch activations.  split mode: those activations are split into bf16 hi/lo planes on the
        # device (q as [T][C], v as [C][T padded to 32]) so the bf16x3 MFMA kernel does the products too.
        st = nb.gn_stats(x, c)
        split = nb.precision == "split"
        q, k, v = nb.buf(c, tm), nb.buf(c, tm), nb.buf(c, tm)
        gam, bet = g[p + "norm.weight"], g[p + "norm.bias"]
        xn = nb.xt_planes(x, c, st, gam, bet, L.ACT_GN, 0.0, 0) if (split and c >= nb.XT_MIN_CO and c % 32 == 0) else -1
        for name, dst, tr in (("q", q, 1 if split else 0), ("k", k, 0), ("v", v, 0 if split else 1)):
            w, b = cw(p + name)
            if xn >= 0:
                nb.conv(xn, dst, c, c, w, b, out_transposed=tr, x_is_planes=True)
            else:
                nb.conv(x, dst, c, c, w, b, stats=st, gamma=gam, beta_gn=bet, in_act=L.ACT_GN, out_transposed=tr)
        nb.release(xn)
        s = nb.buf(0, tm, True)
        tmp = []
        if split:
            qp, vp = nb.buf(c, tm), nb.buf(c, tm, 2)
            nb.split_planes(q, qp, rows=-1, cols=c)                            # q^T [T][C] -> planes
            nb.split_planes(v, vp, rows=c, cols=-1)                            # v [C][T] -> planes, T padded
            tmp = [qp, vp]
            nb.conv(k, s, c, -1, w_buf=qp, w_buf_planes=True, acc_scale=float(int(c) ** (-0.5)))
        else:
            nb.conv(k, s, c, -1, w_buf=q, acc_scale=float(int(c) ** (-0.5)))  # w[b,i,j] = sum_c q[c,i] k[c,j] * C^-0.5
        pT = nb.buf(0, tm, True)
        nb.softmax_t(s, pT)
        a = nb.buf(c, tm)
        if split:
            nb.conv(pT, a, -1, c, w_buf=vp, w_buf_planes=True)
        else:
            nb.conv(pT, a, -1, c, w_buf=v)                                     # h[c,i] = sum_j v^T[j,c] P[i,j]
        out = nb.buf(c, tm)
        w, b = cw(p + "proj_out")
        nb.conv(a, out, c, c, w, b, res=x)
        for t in [st, q, k, v, s, pT, a, x] + tmp:
            nb.release(t)
        return out

    return cw, resblock, attnblock


def build_vae_decoder(ctx: Context, sd: Dict[str, Tensor], scale_factor: float = 1.0, precision: str = "split") -> ConvNet:
    """AutoencoderKL.decode (autoencoder1d.py:55-58) + Decoder1D.forward (:480-512) as an op list.
    The structure (levels, shortcut convs, attention blocks, which level upsamples) is read off the
    key names, exactly what load_state_dict would accept."""
    nb = NetBuilder(ctx.device, precision)
    g = sd
    cw, resblock, attnblock = _vae_block_builders(nb, g)

    zc = g["post_quant_conv.weight"].shape[1]
    h = nb.buf(g["post_quant_conv.weight"].shape[0], 1)
    w, b = cw("post_quant_conv")
    nb.conv(L.BUF_INPUT, h, zc, g["post_quant_conv.weight"].shape[0], w, b, acc_scale=1.0 / float(scale_factor))
    w, b = cw("decoder.conv_in")
    c = g["decoder.conv_in.weight"].shape[0]
    kk = g["decoder.conv_in.weight"].shape[2]
    h2 = nb.buf(c, 1)
    nb.conv(h, h2, g["decoder.conv_in.weight"].shape[1], c, w, b, k=kk, pad=kk // 2)
    nb.release(h)
    h, tm = h2, 1
    h, c = resblock(h, c, tm, "decoder.mid.block_1.")
    h = attnblock(h, c, tm, "decoder.mid.attn_1.")
    h, c = resblock(h, c, tm, "decoder.mid.block_2.")
    levels = sorted({int(k.split(".")[2]) for k in g if k.startswith("decoder.up.")})
    for lvl in reversed(levels):
        nblk = len({int(k.split(".")[4]) for k in g if k.startswith(f"decoder.up.{lvl}.block.")})
        for bi in range(nblk):
            h, c = resblock(h, c, tm, f"decoder.up.{lvl}.block.{bi}.")
            if f"decoder.up.{lvl}.attn.{bi}.norm.weight" in g:
                h = attnblock(h, c, tm, f"decoder.up.{lvl}.attn.{bi}.")
        if f"decoder.up.{lvl}.upsample.conv.weight" in g:
            w, b = cw(f"decoder.up.{lvl}.upsample.conv")
            h2 = nb.buf(c, tm * 2)
            nb.conv(h, h2, c, c, w, b, k=3, pad=1, upsample2=1)
            nb.release(h)
            h, tm = h2, tm * 2
    st = nb.gn_stats(h, c)
    w, b = cw("decoder.conv_out")
    co, kk = g["decoder.conv_out.weight"].shape[0], g["decoder.conv_out.weight"].shape[2]
    nb.conv(h, L.BUF_OUTPUT, c, co, w, b, k=kk, pad=kk // 2, stats=st, gamma=g["decoder.norm_out.weight"],
            beta_gn=g["decoder.norm_out.bias"], in_act=L.ACT_GN_SWISH)
    return ConvNet(ctx, L.NET_VAE, nb, zc, co, tm)


def build_vae_encoder(ctx: Context, sd: Dict[str, Tensor], precision: str = "split") -> ConvNet:
    """AutoencoderKL.encode up to the moments (autoencoder1d.py:49-53: Encoder1D.forward :383-409, then quant_conv :30):
    mel [B,in_ch,T_mel] -> [B, 2*embed_dim, T_mel / 2^len(down_layers)].  Structure read off the key names.
    Downsample1D (:294-313: pad right by one, Conv1d k3 stride 2) runs as two polyphase convolutions on the same kernels:
    out[n] = w0 x[2n] + w2 x[2n+2]  (taps 0 and 2 over the even samples)  +  w1 x[2n+1]  (tap 1 over the odd samples)."""
    nb = NetBuilder(ctx.device, precision)
    g = sd
    cw, resblock, attnblock = _vae_block_builders(nb, g)
    levels = sorted({int(k.split(".")[2]) for k in g if k.startswith("encoder.down.")})
    n_down = sum(1 for lvl in levels if f"encoder.down.{lvl}.downsample.conv.weight" in g)
    tm = 2 ** n_down
    w, b = cw("encoder.conv_in")
    cin, c, kk = g["encoder.conv_in.weight"].shape[1], g["encoder.conv_in.weight"].shape[0], g["encoder.conv_in.weight"].shape[2]
    h = nb.buf(c, tm)
    nb.conv(L.BUF_INPUT, h, cin, c, w, b, k=kk, pad=kk // 2)
    for lvl in levels:
        nblk = len({int(k.split(".")[4]) for k in g if k.startswith(f"encoder.down.{lvl}.block.")})
        for bi in range(nblk):
            h, c = resblock(h, c, tm, f"encoder.down.{lvl}.block.{bi}.")
            if f"encoder.down.{lvl}.attn.{bi}.norm.weight" in g:
                h = attnblock(h, c, tm, f"encoder.down.{lvl}.attn.{bi}.")
        if f"encoder.down.{lvl}.downsample.conv.weight" in g:
            wd, bd = g[f"encoder.down.{lvl}.downsample.conv.weight"].float(), g[f"encoder.down.{lvl}.downsample.conv.bias"]
            assert wd.shape[2] == 3 and tm % 2 == 0
            h2 = nb.buf(c, tm // 2)
            nb.conv(h, h2, c, c, pack.pack_conv(wd[:, :, 0::2].contiguous()), bd, k=2, pad=0, in_stride=2, in_phase=0)
            nb.conv(h, h2, c, c, pack.pack_conv(wd[:, :, 1:2].contiguous()), None, k=1, pad=0, in_stride=2, in_phase=1, beta=1.0)
            nb.release(h)
            h, tm = h2, tm // 2
    h, c = resblock(h, c, tm, "encoder.mid.block_1.")
    h = attnblock(h, c, tm, "encoder.mid.attn_1.")
    h, c = resblock(h, c, tm, "encoder.mid.block_2.")
    st = nb.gn_stats(h, c)
    w, b = cw("encoder.conv_out")
    co, kk = g["encoder.conv_out.weight"].shape[0], g["encoder.conv_out.weight"].shape[2]
    m = nb.buf(co, tm)
    nb.conv(h, m, c, co, w, b, k=kk, pad=kk // 2, stats=st, gamma=g["encoder.norm_out.weight"], beta_gn=g["encoder.norm_out.bias"],
            in_act=L.ACT_GN_SWISH)
    w, b = cw("quant_conv")
    nb.conv(m, L.BUF_OUTPUT, co, g["quant_conv.weight"].shape[0], w, b)
    assert tm == 1
    return ConvNet(ctx, L.NET_VAE_ENCODER, nb, cin, g["quant_conv.weight"].shape[0], 1, in_tmul=2 ** n_down)


def build_hifigan(ctx: Context, sd: Dict[str, Tensor], hp: dict, precision: str = "split", fuse_pairs: Sequence[int] = (32, 64)) -> ConvNet:
    """HifiGanGenerator.forward (vocoder/hifigan/modules/hifigan.py:126-143) as an op list; fully
    driven by the vocoder's config.yaml keys."""
    nb = NetBuilder(ctx.device, precision)
    # exact-fp32 mode: channel counts whose ResBlock1 pairs run fused (respair_f32_kernel); VB_FP32_PAIRS="32,64" / "" for A/B runs
    # (fp32mf: the 32-channel pairs run respair_f32w_kernel - minimal filtering in both convolutions, intermediate in LDS; the 64-channel pairs run
    #  as two unfused minimal-filtering launches: faster than the direct fused pair (1201 -> 1232 mel-s/s) AND than respair_f32w_kernel's 64-channel
    #  form (VB_FP32_PAIRS=32,64: 1196-1198 against 1224; a 2 x 2 wave grid leaves 16 MFMAs per ring step).  VB_MF_PAIRS_OFF=1 = the direct fused
    #  pair kernel.  A/Bs: profiles/r06_mf_pairs_ab.txt, profiles/r06_mf_pair32.txt)
    fp32_pairs = tuple(int(v) for v in os.environ.get("VB_FP32_PAIRS", "32" if precision == "fp32mf" else "32,64").split(",") if v.strip())

    def wt(name):
        if name + ".weight" in sd:
            return sd[name + ".weight"].float()
        return pack.fold_weight_norm(sd[name + ".weight_g"].float(), sd[name + ".weight_v"].float())

    nk = len(hp["resblock_kernel_sizes"])
    c0 = hp["upsample_initial_channel"]
    wpre = wt("conv_pre")
    x = nb.buf(c0, 1)
    nb.conv(L.BUF_INPUT, x, wpre.shape[1], c0, pack.pack_conv(wpre), sd["conv_pre.bias"], k=7, pad=3)
    tm, ch = 1, c0
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        cin, ch = c0 // (2 ** i), c0 // (2 ** (i + 1))
        tm *= u
        xu = nb.buf(ch, tm)
        nb.conv(x, xu, cin, ch, pack.pack_conv_transpose(wt(f"ups.{i}"), u), sd[f"ups.{i}.bias"], in_act=L.ACT_LRELU, in_slope=0.1,
                tr_stride=u, tr_pad=(k - u) // 2, tr_k=k)
        nb.release(x)
        xs = nb.buf(ch, tm)
        for j, (rk, rd) in enumerate(zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"])):
            n = i * nk + j
            r = xu
            for m, d in enumerate(rd):
                last = m == len(rd) - 1
                dst = xs if last else nb.buf(ch, tm)
                al, be = (1.0 / nk, 0.0 if j == 0 else 1.0) if last else (1.0, 0.0)
                # (fp32 pair kernel: 16-byte window DMA needs T % 4 == 0 - guaranteed for every mel length, chunked runs included, when the
                #  stage's length multiplier is a multiple of 4; other stages keep the two unfused launches)
                fuse = (ch in fuse_pairs and (rk - 1) * d <= 64) if precision == "split" else \
                       (ch in fp32_pairs and (rk - 1) * d <= 60 and rk <= 17 and tm % 4 == 0)
                if hp["resblock"] == "1" and fuse and rk % 2 == 1:
                    # narrowest, longest stage: both convolutions of the pair in one launch, intermediate kept in LDS
                    # (measured: 645 us per pair against 2 x 600 us at 32 channels; at 64 channels the fused kernel
                    #  fits one workgroup per CU only and loses, 1525 us against 2 x 640 us - so it is not used there)
                    nb.respair(r, dst, ch, pack.pack_conv(wt(f"resblocks.{n}.convs1.{m}")), sd[f"resblocks.{n}.convs1.{m}.bias"],
                               pack.pack_conv(wt(f"resblocks.{n}.convs2.{m}")), sd[f"resblocks.{n}.convs2.{m}.bias"], rk, d, 0.1, al, be)
                elif hp["resblock"] == "1":
                    t1 = nb.buf(ch, tm)
                    # the LeakyReLU between the two convolutions is applied by the FIRST one's epilogue (t1 has no other reader), not by the
                    # second one's window pass: the same values (max(v, 0.1 v) either way), one in-place LDS pass per 16-channel chunk less
                    # in every second convolution (round 6; VB_LRELU_IN_WINDOW=1 keeps the old op list for the A/B)
                    mid_out = not os.environ.get("VB_LRELU_IN_WINDOW")
                    nb.conv(r, t1, ch, ch, pack.pack_conv(wt(f"resblocks.{n}.convs1.{m}")), sd[f"resblocks.{n}.convs1.{m}.bias"], k=rk,
                            dil=d, pad=(rk * d - d) // 2, in_act=L.ACT_LRELU, in_slope=0.1,
                            out_act=L.ACT_LRELU if mid_out else L.ACT_NONE, out_slope=0.1 if mid_out else 0.0)
                    nb.conv(t1, dst, ch, ch, pack.pack_conv(wt(f"resblocks.{n}.convs2.{m}")), sd[f"resblocks.{n}.convs2.{m}.bias"], k=rk,
                            pad=(rk - 1) // 2, in_act=L.ACT_NONE if mid_out else L.ACT_LRELU, in_slope=0.0 if mid_out else 0.1, res=r, alpha=al,
                            beta=be)
                    nb.release(t1)
                else:
                    nb.conv(r, dst, ch, ch, pack.pack_conv(wt(f"resblocks.{n}.convs.{m}")), sd[f"resblocks.{n}.convs.{m}.bias"], k=rk,
                            dil=d, pad=(rk * d - d) // 2, in_act=L.ACT_LRELU, in_slope=0.1, res=r, alpha=al, beta=be)
                if r != xu:
                    nb.release(r)
                r = dst
        nb.release(xu)
        x = xs
    wpost = wt("conv_post")
    nb.conv(x, L.BUF_OUTPUT, ch, wpost.shape[0], pack.pack_conv(wpost), sd["conv_post.bias"], k=7, pad=3, in_act=L.ACT_LRELU,
            in_slope=0.01, out_act=L.ACT_TANH)
    return ConvNet(ctx, L.NET_VOCODER, nb, wpre.shape[1], wpost.shape[0], tm)


def build_bigvgan(ctx: Context, sd: Dict[str, Tensor], hp: dict, precision: str = "split") -> ConvNet:
    """BigVGAN.forward (vocoder/bigvgan/models.py:181-205) as an op list: conv_pre, per stage ConvTranspose (no activation in
    front of it) + the mean of the AMP blocks (anti-aliased Snake/SnakeBeta -> conv -> ... + x), anti-aliased activation,
    conv_post, tanh.  Driven by the args.yml keys; weights are weight-norm pairs like the HiFi-GAN's."""
    nb = NetBuilder(ctx.device, precision)

    def wt(name):
        if name + ".weight" in sd:
            return sd[name + ".weight"].float()
        return pack.fold_weight_norm(sd[name + ".weight_g"].float(), sd[name + ".weight_v"].float())

    logscale = bool(hp["snake_logscale"])

    def act(name, x, ch, tm):
        out = nb.buf(ch, tm)
        nb.aa_act(x, out, ch, sd[name + ".act.alpha"], sd.get(name + ".act.beta"), logscale)
        return out

    nk = len(hp["resblock_kernel_sizes"])
    c0 = hp["upsample_initial_channel"]
    wpre = wt("conv_pre")
    x = nb.buf(c0, 1)
    nb.conv(L.BUF_INPUT, x, wpre.shape[1], c0, pack.pack_conv(wpre), sd["conv_pre.bias"], k=7, pad=3)
    tm, ch = 1, c0
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        cin, ch = c0 // (2 ** i), c0 // (2 ** (i + 1))
        tm *= u
        xu = nb.buf(ch, tm)
        nb.conv(x, xu, cin, ch, pack.pack_conv_transpose(wt(f"ups.{i}.0"), u), sd[f"ups.{i}.0.bias"], tr_stride=u, tr_pad=(k - u) // 2, tr_k=k)
        nb.release(x)
        xs = nb.buf(ch, tm)
        for j, (rk, rd) in enumerate(zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"])):
            n = i * nk + j
            r = xu
            for m, d in enumerate(rd):
                last = m == len(rd) - 1
                dst = xs if last else nb.buf(ch, tm)
                al, be = (1.0 / nk, 0.0 if j == 0 else 1.0) if last else (1.0, 0.0)
                if hp["resblock"] == "1":
                    a1 = act(f"resblocks.{n}.activations.{2 * m}", r, ch, tm)
                    t1 = nb.buf(ch, tm)
                    nb.conv(a1, t1, ch, ch, pack.pack_conv(wt(f"resblocks.{n}.convs1.{m}")), sd[f"resblocks.{n}.convs1.{m}.bias"], k=rk,
                            dil=d, pad=(rk * d - d) // 2)
                    nb.release(a1)
                    a2 = act(f"resblocks.{n}.activations.{2 * m + 1}", t1, ch, tm)
                    nb.release(t1)
                    nb.conv(a2, dst, ch, ch, pack.pack_conv(wt(f"resblocks.{n}.convs2.{m}")), sd[f"resblocks.{n}.convs2.{m}.bias"], k=rk,
                            pad=(rk - 1) // 2, res=r, alpha=al, beta=be)
                    nb.release(a2)
                else:
                    a1 = act(f"resblocks.{n}.activations.{m}", r, ch, tm)
                    nb.conv(a1, dst, ch, ch, pack.pack_conv(wt(f"resblocks.{n}.convs.{m}")), sd[f"resblocks.{n}.convs.{m}.bias"], k=rk,
                            dil=d, pad=(rk * d - d) // 2, res=r, alpha=al, beta=be)
                    nb.release(a1)
                if r != xu:
                    nb.release(r)
                r = dst
        nb.release(xu)
        x = xs
    ap = act("activation_post", x, ch, tm)
    wpost = wt("conv_post")
    nb.conv(ap, L.BUF_OUTPUT, ch, wpost.shape[0], pack.pack_conv(wpost), sd["conv_post.bias"], k=7, pad=3, out_act=L.ACT_TANH)
    return ConvNet(ctx, L.NET_VOCODER, nb, wpre.shape[1], wpost.shape[0], tm)
