"""CPU oracle: a plain-PyTorch fp32 restatement of the AccompBand inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``versband_amd/`` imports this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may call it, and only as the checker / the timed CPU baseline.

Pinned against the real reference: ``oracle/gen_golden.py`` imports
``/root/reference`` (build container only), loads the same synthetic weights,
injects the same noise and stores the reference's outputs in
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those fixtures.  The one piece with no reference source in the tree
is the fixed-step Euler of ``torchdyn.NeuralODE`` (un-pinned dependency, SURVEY
§8 A4): **parity unpinned** for that solver loop alone - it is restated from its
published algorithm, x <- x + (t[k+1]-t[k]) * f(t[k], x).
The mel front-end adds a second such piece: the mel FILTERBANK is
``librosa.filters.mel`` upstream (librosa==0.10.1, absent here) - restated from its
published algorithm, **parity unpinned** for ``slaney_mel_filterbank`` alone; the
rest of ``melnet_forward`` is pinned against the reference's own MelNet.

All citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ---------------------------------------------------------------------------
# small pieces
# ---------------------------------------------------------------------------


def rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """RMSNorm  ldm/modules/diffusionmodules/flag_large_dit_moe.py:52-77."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)) * w


def modulate(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    """flag_large_dit_moe.py:80-81."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def timestep_embedding(t: Tensor, dim: int = 256, max_period: float = 10000.0) -> Tensor:
    """TimestepEmbedder.timestep_embedding  flag_large_dit_moe.py:110-128."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def rope_table(head_dim: int, end: int, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """precompute_freqs_cis  vocal2music_moe.py:436-475 -> (cos, sin) [end, head_dim/2]."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    t = torch.arange(end, dtype=torch.float32)
    ang = torch.outer(t, freqs).float()
    return torch.cos(ang), torch.sin(ang)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """Attention.apply_rotary_emb  flag_large_dit_moe.py:237-269.
    x [B,T,H,hd]; rotates adjacent pairs (2j, 2j+1) by the angle of position t."""
    B, T, H, hd = x.shape
    xr = x.float().reshape(B, T, H, hd // 2, 2)
    a, b = xr[..., 0], xr[..., 1]
    c = cos[:T].view(1, T, 1, hd // 2)
    s = sin[:T].view(1, T, 1, hd // 2)
    out = torch.stack([a * c - b * s, a * s + b * c], dim=-1)
    return out.flatten(3)


def sdpa(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(d)) v with q,k,v [B,H,S,d] (no mask: all-ones masks,
    vocal2music_moe.py:403-404)."""
    d = q.shape[-1]
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)
    return torch.matmul(torch.softmax(s, dim=-1), v)


def swiglu(x: Tensor, w1: Tensor, w2: Tensor, w3: Tensor) -> Tensor:
    """FeedForward.forward  flag_large_dit_moe.py:480-485."""
    return F.linear(F.silu(F.linear(x, w1)) * F.linear(x, w3), w2)


def gumbel_from_exponential(e: Tensor) -> Tensor:
    """-log(Exp(1))  vocal2music_moe.py:82."""
    return -e.log()


def router_top1(logits: Tensor, gumbel: Tensor, temperature: float = 2.0) -> Tuple[Tensor, Tensor]:
    """Hard Gumbel-softmax selection  vocal2music_moe.py:81-93 (hard=True at inference).

    Returns (index int64 [N], straight-through weight of the chosen expert [N]).
    Index = first maximum of softmax((logits+g)/temperature), like torch.max."""
    y = ((logits + gumbel) / temperature).softmax(dim=-1)
    idx = y.max(dim=-1, keepdim=True)[1]
    p = y.gather(-1, idx)
    w = (1.0 - p) + p
    return idx.squeeze(-1), w.squeeze(-1)


# ---------------------------------------------------------------------------
# DiT + Band-MoE
# ---------------------------------------------------------------------------


class DiTDims:
    def __init__(self, sd: Dict[str, Tensor], prefix: str = ""):
        self.D = sd[prefix + "proj_in.weight"].shape[0]
        self.C = sd[prefix + "proj_in.weight"].shape[1]
        self.heads = sd[prefix + "blocks.0.attention.gate"].shape[0]
        self.hd = self.D // self.heads
        self.E = sd[prefix + "blocks.0.feed_forward.caption_gating_network.weight"].shape[0]
        self.depth = 1 + max(int(k[len(prefix):].split(".")[1]) for k in sd if k.startswith(prefix + "blocks."))
        self.eps = 1e-5


def dit_precompute(sd: Dict[str, Tensor], t5: Tensor, midi: Tensor, beats: Tensor, T: int) -> Dict[str, Tensor]:
    """Everything in TxtFlagLargeDiT.forward that does not depend on (x, t):
    vocal2music_moe.py:384-393 (acoustic stem), :407-413 (caption embedding + pooled
    embedding), and per block the context K/V of Attention (flag_large_dit_moe.py:389-390),
    the K/V of MoE.cross_attention and the acoustic gate logits (vocal2music_moe.py:119,142).

    t5 [B,L,ori] float; midi/beats [B,1,2T] int64."""
    dm = DiTDims(sd)
    D, H, hd = dm.D, dm.heads, dm.hd
    B, L, _ = t5.shape
    m = F.embedding(midi.squeeze(1), sd["midi_embedding.weight"]).transpose(1, 2)
    b = F.embedding(beats.squeeze(1), sd["beats_embedding.weight"]).transpose(1, 2)
    m = F.avg_pool1d(F.leaky_relu(F.conv1d(m, sd["midi_proj.0.weight"], sd["midi_proj.0.bias"], padding=2), 0.01), 2)
    b = F.avg_pool1d(F.leaky_relu(F.conv1d(b, sd["beats_proj.0.weight"], sd["beats_proj.0.bias"], padding=2), 0.01), 2)
    ac = F.conv1d(m + b, sd["final_proj.weight"], sd["final_proj.bias"]).transpose(1, 2)  # [B,T_ac,D]
    if abs(T - ac.shape[1]) <= 2:                                       # vocal2music_moe.py:397-401
        if T > ac.shape[1]:
            ac = torch.cat([ac, ac[:, -1:, :].repeat(1, T - ac.shape[1], 1)], dim=1)
        else:
            ac = ac[:, :T, :]
    cap = F.linear(t5, sd["c_embedder.mlp.0.weight"], sd["c_embedder.mlp.0.bias"])
    cap = F.gelu(cap)
    cap = F.linear(cap, sd["c_embedder.mlp.2.weight"], sd["c_embedder.mlp.2.bias"])
    cap = F.layer_norm(cap, (D,), sd["c_embedder.mlp.3.weight"], sd["c_embedder.mlp.3.bias"], 1e-5)
    pooled = cap.sum(dim=1) / float(L)
    cemb = F.linear(F.layer_norm(pooled, (D,), sd["cap_embedder.0.weight"], sd["cap_embedder.0.bias"], 1e-5),
                    sd["cap_embedder.1.weight"], sd["cap_embedder.1.bias"])
    out = {"acoustic": ac.contiguous(), "caption": cap.contiguous(), "cap_emb": cemb.contiguous()}
    for i in range(dm.depth):
        p = f"blocks.{i}."
        y = rmsnorm(cap, sd[p + "attention_y_norm.weight"], dm.eps)
        out[f"ky{i}"] = F.linear(y, sd[p + "attention.wk_y.weight"]).view(B, L, H, hd)
        out[f"vy{i}"] = F.linear(y, sd[p + "attention.wv_y.weight"]).view(B, L, H, hd)
        w_in = sd[p + "feed_forward.cross_attention.in_proj_weight"]
        b_in = sd[p + "feed_forward.cross_attention.in_proj_bias"]
        out[f"kc{i}"] = F.linear(cap, w_in[D:2 * D], b_in[D:2 * D]).view(B, L, H, hd)
        out[f"vc{i}"] = F.linear(cap, w_in[2 * D:], b_in[2 * D:]).view(B, L, H, hd)
        out[f"la{i}"] = F.linear(ac, sd[p + "feed_forward.acoustic_gating_network.weight"],
                                 sd[p + "feed_forward.acoustic_gating_network.bias"])       # [B,T,E]
    return out


def dit_forward(sd: Dict[str, Tensor], x: Tensor, t_idx: Tensor, cond: Dict[str, Tensor],
                noise: Sequence[Tuple[Tensor, Tensor, Tensor]], dense: bool = False,
                temperature: float = 2.0, return_aux: bool = False):
    """TxtFlagLargeDiT.forward  vocal2music_moe.py:375-434 (per-NFE part).

    x [B,C,T] f32; t_idx [B] int64; cond from ``dit_precompute``;
    noise[i] = (E1 [N,2], E2 [N,E], E3 [N,E]) Exp(1) draws of block i in the order
    MoE.forward consumes them (vocal2music_moe.py:134,150,151).
    dense=True evaluates all 3E expert FFNs on every token like the reference does
    (:157-178); dense=False is the routed + band-sliced evaluation."""
    dm = DiTDims(sd)
    D, H, hd, E = dm.D, dm.heads, dm.hd, dm.E
    B, C, T = x.shape
    N = B * T
    cos, sin = rope_table(hd, T)
    temb = F.linear(timestep_embedding(t_idx), sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])
    temb = F.linear(F.silu(temb), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])
    c = temb + cond["cap_emb"]
    ac = cond["acoustic"]
    h = F.conv1d(x, sd["proj_in.weight"], sd["proj_in.bias"], padding=2).transpose(1, 2)
    h = ac + h
    aux: Dict[str, Tensor] = {}
    for i in range(dm.depth):
        p = f"blocks.{i}."
        mod = F.linear(F.silu(c), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"])
        sh1, sc1, g1, sh2, sc2, g2 = mod.chunk(6, dim=1)
        # --- Attention.forward flag_large_dit_moe.py:323-406
        u = modulate(rmsnorm(h, sd[p + "attention_norm.weight"], dm.eps), sh1, sc1)
        q = F.linear(u, sd[p + "attention.wq.weight"]).view(B, T, H, hd)
        k = F.linear(u, sd[p + "attention.wk.weight"]).view(B, T, H, hd)
        v = F.linear(u, sd[p + "attention.wv.weight"]).view(B, T, H, hd)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        qh = q.permute(0, 2, 1, 3)
        a_self = sdpa(qh, k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3))
        a_cross = sdpa(qh, cond[f"ky{i}"].permute(0, 2, 1, 3), cond[f"vy{i}"].permute(0, 2, 1, 3))
        gate = sd[p + "attention.gate"].tanh().view(1, H, 1, 1)
        a = (a_self + a_cross * gate).permute(0, 2, 1, 3).reshape(B, T, D)
        h = h + g1.unsqueeze(1) * F.linear(a, sd[p + "attention.wo.weight"])
        # --- MoE.forward vocal2music_moe.py:117-185
        u = modulate(rmsnorm(h, sd[p + "ffn_norm.weight"], dm.eps), sh2, sc2)
        w_in = sd[p + "feed_forward.cross_attention.in_proj_weight"]
        b_in = sd[p + "feed_forward.cross_attention.in_proj_bias"]
        qm = F.linear(u, w_in[:D], b_in[:D]).view(B, T, H, hd).permute(0, 2, 1, 3)
        cq = sdpa(qm, cond[f"kc{i}"].permute(0, 2, 1, 3), cond[f"vc{i}"].permute(0, 2, 1, 3))
        cq = cq.permute(0, 2, 1, 3).reshape(B, T, D)
        cq = F.linear(cq, sd[p + "feed_forward.cross_attention.out_proj.weight"],
                      sd[p + "feed_forward.cross_attention.out_proj.bias"]).reshape(N, D)
        un = u.reshape(N, D)
        e1, e2, e3 = noise[i]
        hl = F.linear(temb, sd[p + "feed_forward.high_level_gating_network.weight"],
                      sd[p + "feed_forward.high_level_gating_network.bias"]).repeat_interleave(T, dim=0)
        hp = ((hl + gumbel_from_exponential(e1)) / 1.0).softmax(dim=-1)
        m_c, m_a = hp[:, 0:1], hp[:, 1:2]
        lc = F.linear(cq, sd[p + "feed_forward.caption_gating_network.weight"],
                      sd[p + "feed_forward.caption_gating_network.bias"])
        la = cond[f"la{i}"].reshape(N, E)
        ic, wc = router_top1(lc, gumbel_from_exponential(e2), temperature)
        ia, wa = router_top1(la, gumbel_from_exponential(e3), temperature)
        y = torch.zeros_like(un)
        if dense:
            zc = torch.zeros_like(un)
            za = torch.zeros_like(un)
            for e in range(E):
                q_ = f"{p}feed_forward.caption_experts.{e}."
                w = torch.where(ic == e, wc, torch.zeros_like(wc)).unsqueeze(1)
                zc = zc + swiglu(un, sd[q_ + "w1.weight"], sd[q_ + "w2.weight"], sd[q_ + "w3.weight"]) * w * m_c
            for e in range(E):
                q_ = f"{p}feed_forward.acoustic_experts.{e}."
                w = torch.where(ia == e, wa, torch.zeros_like(wa)).unsqueeze(1)
                za = za + swiglu(un, sd[q_ + "w1.weight"], sd[q_ + "w2.weight"], sd[q_ + "w3.weight"]) * w * m_a
            y = zc + za
        else:
            zc = torch.zeros_like(un)
            za = torch.zeros_like(un)
            for e in range(E):
                sel = (ic == e).nonzero().squeeze(1)
                if sel.numel():
                    q_ = f"{p}feed_forward.caption_experts.{e}."
                    o = swiglu(un[sel], sd[q_ + "w1.weight"], sd[q_ + "w2.weight"], sd[q_ + "w3.weight"])
                    zc[sel] = o * wc[sel].unsqueeze(1) * m_c[sel]
                sel = (ia == e).nonzero().squeeze(1)
                if sel.numel():
                    q_ = f"{p}feed_forward.acoustic_experts.{e}."
                    o = swiglu(un[sel], sd[q_ + "w1.weight"], sd[q_ + "w2.weight"], sd[q_ + "w3.weight"])
                    za[sel] = o * wa[sel].unsqueeze(1) * m_a[sel]
            y = zc + za
        z = torch.zeros_like(y)
        band = D // E
        for e in range(E):
            q_ = f"{p}feed_forward.freq_experts.{e}."
            lo, hi = band * e, band * (e + 1)
            if dense:
                region = torch.zeros_like(y)
                region[:, lo:hi] = 1.0
                z[:, lo:hi] = swiglu(y * region, sd[q_ + "w1.weight"], sd[q_ + "w2.weight"], sd[q_ + "w3.weight"])[:, lo:hi]
            else:
                z[:, lo:hi] = swiglu(y[:, lo:hi], sd[q_ + "w1.weight"][:, lo:hi], sd[q_ + "w2.weight"][lo:hi, :],
                                     sd[q_ + "w3.weight"][:, lo:hi])
        h = h + g2.unsqueeze(1) * z.view(B, T, D)
        if return_aux:
            aux[f"ic{i}"], aux[f"ia{i}"] = ic, ia
            aux[f"hp{i}"] = hp
            aux[f"h{i}"] = h.clone()
            aux[f"lc{i}"] = lc
    modf = F.linear(F.silu(c), sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"])
    shift, scale = modf.chunk(2, dim=1)
    o = modulate(F.layer_norm(h, (D,), None, None, 1e-6), shift, scale)
    o = F.linear(o, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"]).transpose(1, 2).contiguous()
    if return_aux:
        return o, aux
    return o


# ---------------------------------------------------------------------------
# flow-matching sampler (CFG + fixed-step Euler)
# ---------------------------------------------------------------------------


def t_index_table(timesteps: int, t_start: Optional[int] = None) -> Tuple[Tensor, List[int]]:
    """t_span and the integer diffusion indices the reference feeds the DiT:
    t_span = linspace(0,1,timesteps) (cfm1_audio_sampler.py:108); per Euler step k the
    solver passes t = t_span[k] and Wrapper_cfg truncates float32(t)*1000 to long
    (cfm1_audio.py:156)."""
    t_span = torch.linspace(0, 1, timesteps)
    if t_start is not None:
        t_span = t_span[t_start:]
    idx = [int(torch.tensor([t_span[k] * 1000]).long().item()) for k in range(len(t_span) - 1)]
    return t_span, idx


def sample_cfg(sd: Dict[str, Tensor], x0: Tensor, cond_c: Dict[str, Tensor], cond_u: Optional[Dict[str, Tensor]],
               scale: float, timesteps: int, noise_fn, dense: bool = False, t_start: Optional[int] = None,
               return_traj: bool = False):
    """CFMSampler.sample_cfg (cfm1_audio_sampler.py:87-116) + Wrapper_cfg.forward
    (cfm1_audio.py:154-162) + torchdyn fixed-step Euler (restated, parity unpinned).

    noise_fn(step, branch) -> per-block noise list; branch 0 = cond, 1 = uncond (the
    reference evaluates cond first, cfm1_audio.py:158-159)."""
    t_span, idx = t_index_table(timesteps, t_start)
    x = x0.clone()
    B = x.shape[0]
    traj = [x.clone()]
    t = t_span[0]
    for k in range(len(t_span) - 1):
        dt = t_span[k + 1] - t
        ti = torch.full((B,), idx[k], dtype=torch.long)
        e_c = dit_forward(sd, x, ti, cond_c, noise_fn(k, 0), dense=dense)
        if cond_u is not None:
            e_u = dit_forward(sd, x, ti, cond_u, noise_fn(k, 1), dense=dense)
            e = e_u + scale * (e_c - e_u)
        else:
            e = e_c
        x = x + dt * e
        t = t + dt
        if return_traj:
            traj.append(x.clone())
    if return_traj:
        return x, torch.stack(traj)
    return x


# ---------------------------------------------------------------------------
# VAE decoder
# ---------------------------------------------------------------------------


def _gn_swish(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    h = F.group_norm(x, 32, w, b, 1e-6)          # Normalize autoencoder1d.py:165-166
    return h * torch.sigmoid(h)                  # nonlinearity :168-170


def _resblock(sd, p: str, x: Tensor) -> Tensor:
    """ResnetBlock1D.forward  autoencoder1d.py:210-231 (temb is None, dropout 0)."""
    h = _gn_swish(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    w = sd[p + "conv1.weight"]
    h = F.conv1d(h, w, sd[p + "conv1.bias"], padding=w.shape[2] // 2)
    h = _gn_swish(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    w = sd[p + "conv2.weight"]
    h = F.conv1d(h, w, sd[p + "conv2.bias"], padding=w.shape[2] // 2)
    if (p + "nin_shortcut.weight") in sd:
        x = F.conv1d(x, sd[p + "nin_shortcut.weight"], sd[p + "nin_shortcut.bias"])
    return x + h


def _attnblock(sd, p: str, x: Tensor) -> Tensor:
    """AttnBlock1D.forward  autoencoder1d.py:254-274.  Note the reference unpacks
    ``b,t,c = q.shape`` on a [B,C,T] tensor, so the softmax scale is C**-0.5."""
    h = F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    q = F.conv1d(h, sd[p + "q.weight"], sd[p + "q.bias"])
    k = F.conv1d(h, sd[p + "k.weight"], sd[p + "k.bias"])
    v = F.conv1d(h, sd[p + "v.weight"], sd[p + "v.bias"])
    C = q.shape[1]
    w = torch.bmm(q.permute(0, 2, 1), k) * (int(C) ** (-0.5))
    w = torch.softmax(w, dim=2)
    h = torch.bmm(v, w.permute(0, 2, 1))
    h = F.conv1d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return x + h


def vae_decode(sd: Dict[str, Tensor], z: Tensor, scale_factor: float = 1.0, prefix: str = "") -> Tensor:
    """decode_first_stage (ddpm_audio.py:379-392) -> AutoencoderKL.decode
    (autoencoder1d.py:55-58) -> Decoder1D.forward (:480-512).  Structure (levels, which
    level upsamples, where attention sits) is derived from the key names present."""
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    z = z * (1.0 / scale_factor)
    h = F.conv1d(z, g["post_quant_conv.weight"], g["post_quant_conv.bias"])
    w = g["decoder.conv_in.weight"]
    h = F.conv1d(h, w, g["decoder.conv_in.bias"], padding=w.shape[2] // 2)
    h = _resblock(g, "decoder.mid.block_1.", h)
    h = _attnblock(g, "decoder.mid.attn_1.", h)
    h = _resblock(g, "decoder.mid.block_2.", h)
    levels = sorted({int(k.split(".")[2]) for k in g if k.startswith("decoder.up.")})
    for lvl in reversed(levels):
        nb = len({int(k.split(".")[4]) for k in g if k.startswith(f"decoder.up.{lvl}.block.")})
        for b in range(nb):
            h = _resblock(g, f"decoder.up.{lvl}.block.{b}.", h)
            if f"decoder.up.{lvl}.attn.{b}.norm.weight" in g:
                h = _attnblock(g, f"decoder.up.{lvl}.attn.{b}.", h)
        if f"decoder.up.{lvl}.upsample.conv.weight" in g:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")          # Upsample1D :287-291
            h = F.conv1d(h, g[f"decoder.up.{lvl}.upsample.conv.weight"], g[f"decoder.up.{lvl}.upsample.conv.bias"], padding=1)
    h = _gn_swish(h, g["decoder.norm_out.weight"], g["decoder.norm_out.bias"])
    w = g["decoder.conv_out.weight"]
    return F.conv1d(h, w, g["decoder.conv_out.bias"], padding=w.shape[2] // 2)


def vae_encode(sd: Dict[str, Tensor], x: Tensor, prefix: str = "") -> Tensor:
    """AutoencoderKL.encode up to the posterior parameters (autoencoder1d.py:49-53): Encoder1D.forward (:383-409) then
    quant_conv (:30).  x [B,80,T_mel] -> moments [B, 2*embed, T_mel/2^downs]; mean = first half of the channels."""
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    w = g["encoder.conv_in.weight"]
    h = F.conv1d(x, w, g["encoder.conv_in.bias"], padding=w.shape[2] // 2)
    levels = sorted({int(k.split(".")[2]) for k in g if k.startswith("encoder.down.")})
    for lvl in levels:
        nb = len({int(k.split(".")[4]) for k in g if k.startswith(f"encoder.down.{lvl}.block.")})
        for b in range(nb):
            h = _resblock(g, f"encoder.down.{lvl}.block.{b}.", h)
            if f"encoder.down.{lvl}.attn.{b}.norm.weight" in g:
                h = _attnblock(g, f"encoder.down.{lvl}.attn.{b}.", h)
        if f"encoder.down.{lvl}.downsample.conv.weight" in g:                 # Downsample1D :294-313
            h = F.pad(h, (0, 1), mode="constant", value=0)
            h = F.conv1d(h, g[f"encoder.down.{lvl}.downsample.conv.weight"], g[f"encoder.down.{lvl}.downsample.conv.bias"], stride=2)
    h = _resblock(g, "encoder.mid.block_1.", h)
    h = _attnblock(g, "encoder.mid.attn_1.", h)
    h = _resblock(g, "encoder.mid.block_2.", h)
    h = _gn_swish(h, g["encoder.norm_out.weight"], g["encoder.norm_out.bias"])
    w = g["encoder.conv_out.weight"]
    h = F.conv1d(h, w, g["encoder.conv_out.bias"], padding=w.shape[2] // 2)
    return F.conv1d(h, g["quant_conv.weight"], g["quant_conv.bias"])


def gaussian_posterior(moments: Tensor, noise: Tensor = None, scale_factor: float = 1.0) -> Tensor:
    """DiagonalGaussianDistribution (ldm/modules/distributions/distributions.py:4-27) + get_first_stage_encoding
    (ddpm_audio.py:163-170): logvar clamped to [-30, 20]; sample = mean + exp(0.5 logvar) * noise (mode when noise is None)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
    return scale_factor * z


# ---------------------------------------------------------------------------
# T5 text encoder - the reference calls transformers.T5EncoderModel (ldm/modules/encoders/modules.py:197-221);
# transformers is a third-party dependency that requirements.txt does not pin.  Restated from the published HF
# implementation (modeling_t5.py: T5LayerNorm, T5Attention incl. _relative_position_bucket, T5DenseGatedActDense, T5Stack)
# and pinned against the transformers installed in the build container (tests/golden/t5_encode*.npz).
# ---------------------------------------------------------------------------


def t5_relative_position_bucket(relative_position: Tensor, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """T5Attention._relative_position_bucket with bidirectional=True (encoder)."""
    num_buckets //= 2
    rb = (relative_position > 0).to(torch.long) * num_buckets
    rp = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return rb + torch.where(is_small, rp, large)


def t5_position_bias(rel_emb: Tensor, L: int, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """T5Attention.compute_bias: [heads, L, L] = embedding[bucket(key - query)]."""
    ctx = torch.arange(L, dtype=torch.long)[:, None]
    mem = torch.arange(L, dtype=torch.long)[None, :]
    bucket = t5_relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return rel_emb[bucket].permute(2, 0, 1).contiguous()


def _t5_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    return w * (x * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps))


def t5_encode(sd: Dict[str, Tensor], ids: Tensor, num_heads: int = 16, eps: float = 1e-6, num_buckets: int = 32,
              max_distance: int = 128) -> Tensor:
    """T5EncoderModel(input_ids=ids).last_hidden_state, no attention mask, dropout off.  ids [B,L] -> [B,L,d_model]."""
    x = sd["shared.weight"][ids]
    B, L, D = x.shape
    n_layers = len({int(k.split(".")[2]) for k in sd if k.startswith("encoder.block.")})
    bias = t5_position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], L, num_buckets, max_distance)
    for i in range(n_layers):
        p = f"encoder.block.{i}."
        n = _t5_norm(x, sd[p + "layer.0.layer_norm.weight"], eps)
        q = (n @ sd[p + "layer.0.SelfAttention.q.weight"].t()).view(B, L, num_heads, -1).transpose(1, 2)
        k = (n @ sd[p + "layer.0.SelfAttention.k.weight"].t()).view(B, L, num_heads, -1).transpose(1, 2)
        v = (n @ sd[p + "layer.0.SelfAttention.v.weight"].t()).view(B, L, num_heads, -1).transpose(1, 2)
        sc = torch.matmul(q, k.transpose(3, 2)) + bias[None]                      # no 1/sqrt(d) in T5
        a = torch.matmul(torch.softmax(sc.float(), dim=-1), v).transpose(1, 2).reshape(B, L, -1)
        x = x + a @ sd[p + "layer.0.SelfAttention.o.weight"].t()
        n = _t5_norm(x, sd[p + "layer.1.layer_norm.weight"], eps)
        g = n @ sd[p + "layer.1.DenseReluDense.wi_0.weight"].t()
        g = 0.5 * g * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (g + 0.044715 * torch.pow(g, 3.0))))     # NewGELUActivation
        x = x + (g * (n @ sd[p + "layer.1.DenseReluDense.wi_1.weight"].t())) @ sd[p + "layer.1.DenseReluDense.wo.weight"].t()
    return _t5_norm(x, sd["encoder.final_layer_norm.weight"], eps)


# ---------------------------------------------------------------------------
# HiFi-GAN generator
# ---------------------------------------------------------------------------


def fold_weight_norm(g: Tensor, v: Tensor) -> Tensor:
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v|| over all dims but 0."""
    n = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def _hg_w(sd, name: str) -> Tensor:
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    return fold_weight_norm(sd[name + ".weight_g"], sd[name + ".weight_v"])


def hifigan_forward(sd: Dict[str, Tensor], hp: dict, mel: Tensor) -> Tensor:
    """HifiGanGenerator.forward  vocoder/hifigan/modules/hifigan.py:126-143.
    mel [B,80,T] -> wav [B,1,T*hop].  hp = the vocoder config dict."""
    nk = len(hp["resblock_kernel_sizes"])
    x = F.conv1d(mel, _hg_w(sd, "conv_pre"), sd["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, _hg_w(sd, f"ups.{i}"), sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"])):
            n = i * nk + j
            r = x
            if hp["resblock"] == "1":                                   # ResBlock1 :27-64
                for m, d in enumerate(rd):
                    xt = F.leaky_relu(r, 0.1)
                    xt = F.conv1d(xt, _hg_w(sd, f"resblocks.{n}.convs1.{m}"), sd[f"resblocks.{n}.convs1.{m}.bias"],
                                  dilation=d, padding=int((rk * d - d) / 2))
                    xt = F.leaky_relu(xt, 0.1)
                    xt = F.conv1d(xt, _hg_w(sd, f"resblocks.{n}.convs2.{m}"), sd[f"resblocks.{n}.convs2.{m}.bias"],
                                  padding=int((rk - 1) / 2))
                    r = xt + r
            else:                                                       # ResBlock2 :67-88
                for m, d in enumerate(rd):
                    xt = F.leaky_relu(r, 0.1)
                    xt = F.conv1d(xt, _hg_w(sd, f"resblocks.{n}.convs.{m}"), sd[f"resblocks.{n}.convs.{m}.bias"],
                                  dilation=d, padding=int((rk * d - d) / 2))
                    r = xt + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)                                                 # default slope 0.01 (:139)
    x = F.conv1d(x, _hg_w(sd, "conv_post"), sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def normalize_loudness(wav, target_loudness: float):
    """scripts/test_final.py:342-347."""
    import numpy as np
    rms = np.sqrt(np.mean(wav ** 2))
    gain = target_loudness - 20 * np.log10(rms)
    return wav * 10 ** (gain / 20)


# ---------------------------------------------------------------------------
# BigVGAN generator: vocoder/bigvgan/models.py:133-213, activations.py, alias_free_torch/{act,resample,filter}.py
# ---------------------------------------------------------------------------


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> Tensor:
    """alias_free_torch/filter.py:28-57 -> [kernel_size]."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if even else torch.arange(kernel_size) - half_size
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return f / f.sum()


def aa_activation(x: Tensor, alpha: Tensor, beta: Optional[Tensor], logscale: bool, ratio: int = 2, k: int = 12) -> Tensor:
    """Activation1d (act.py): UpSample1d(2, 12) -> Snake / SnakeBeta -> DownSample1d(2, 12) (resample.py:10-49)."""
    C = x.shape[1]
    f = kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, k).view(1, 1, k).expand(C, -1, -1)
    pad = k // ratio - 1
    pad_left = pad * ratio + (k - ratio) // 2
    pad_right = pad * ratio + (k - ratio + 1) // 2
    y = F.pad(x, (pad, pad), mode="replicate")
    y = ratio * F.conv_transpose1d(y, f, stride=ratio, groups=C)
    y = y[..., pad_left:-pad_right]
    a = alpha[None, :, None]
    b = a if beta is None else beta[None, :, None]
    if logscale:
        a, b = torch.exp(a), torch.exp(b)
    y = y + (1.0 / (b + 1e-9)) * torch.pow(torch.sin(y * a), 2)
    y = F.pad(y, (k // 2 - 1, k // 2), mode="replicate")            # LowPassFilter1d: even kernel -> pad_left = k/2 - 1
    return F.conv1d(y, f, stride=ratio, groups=C)


def bigvgan_forward(sd: Dict[str, Tensor], hp: dict, mel: Tensor) -> Tensor:
    """BigVGAN.forward models.py:181-205.  mel [B,num_mels,T] -> wav [B,1,T*hop]."""
    nk = len(hp["resblock_kernel_sizes"])
    logscale = bool(hp["snake_logscale"])

    def act(name, x):
        return aa_activation(x, sd[name + ".act.alpha"], sd.get(name + ".act.beta"), logscale)

    x = F.conv1d(mel, _hg_w(sd, "conv_pre"), sd["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, _hg_w(sd, f"ups.{i}.0"), sd[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)     # no activation before it
        xs = None
        for j, (rk, rd) in enumerate(zip(hp["resblock_kernel_sizes"], hp["resblock_dilation_sizes"])):
            n = i * nk + j
            r = x
            if hp["resblock"] == "1":                                   # AMPBlock1 :29-84
                for m, d in enumerate(rd):
                    xt = act(f"resblocks.{n}.activations.{2 * m}", r)
                    xt = F.conv1d(xt, _hg_w(sd, f"resblocks.{n}.convs1.{m}"), sd[f"resblocks.{n}.convs1.{m}.bias"], dilation=d,
                                  padding=int((rk * d - d) / 2))
                    xt = act(f"resblocks.{n}.activations.{2 * m + 1}", xt)
                    xt = F.conv1d(xt, _hg_w(sd, f"resblocks.{n}.convs2.{m}"), sd[f"resblocks.{n}.convs2.{m}.bias"], padding=int((rk - 1) / 2))
                    r = xt + r
            else:                                                       # AMPBlock2 :87-130
                for m, d in enumerate(rd):
                    xt = act(f"resblocks.{n}.activations.{m}", r)
                    xt = F.conv1d(xt, _hg_w(sd, f"resblocks.{n}.convs.{m}"), sd[f"resblocks.{n}.convs.{m}.bias"], dilation=d,
                                  padding=int((rk * d - d) / 2))
                    r = xt + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = act("activation_post", x)
    x = F.conv1d(x, _hg_w(sd, "conv_post"), sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


# ---------------------------------------------------------------------------
# Log-mel front-end: preprocess/NAT_mel.py:42-86 (MelNet)
# ---------------------------------------------------------------------------


def slaney_mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: Optional[float]):
    """What NAT_mel.py:53 asks of `librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=)` (librosa==0.10.1, requirements.txt:3;
    NOT in this image - restated from the published algorithm of its defaults htk=False, norm='slaney'; **parity unpinned** for
    this function alone).  Scalar loops on purpose: the product's numpy version (versband_amd/melnet.py) is written independently."""
    import numpy as np
    f_sp, brk, step = 200.0 / 3.0, 1000.0, math.log(6.4) / 27.0

    def to_mel(f):
        return f / f_sp if f < brk else brk / f_sp + math.log(f / brk) / step

    def to_hz(m):
        return f_sp * m if m < brk / f_sp else brk * math.exp(step * (m - brk / f_sp))

    fmax = sr / 2.0 if fmax is None else float(fmax)
    lo, hi = to_mel(float(fmin)), to_mel(fmax)
    edges = [to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    nb = n_fft // 2 + 1
    fb = np.zeros((n_mels, nb), dtype=np.float64)
    for i in range(n_mels):
        l, c, r = edges[i], edges[i + 1], edges[i + 2]
        for k in range(nb):
            f = k * (sr / 2.0) / (nb - 1)
            v = min((f - l) / (c - l), (r - f) / (r - c))
            if v > 0.0:
                fb[i, k] = v * 2.0 / (r - l)
    return fb.astype(np.float32)


def melnet_forward(wav: Tensor, hp: dict, mel_basis: Tensor, center: bool = False) -> Tensor:
    """MelNet.forward (NAT_mel.py:64-86), complex=False branch: wav [B, L] -> log10-mel [B, n_mels, frames].
    The STFT is restated as an explicit windowed rFFT of strided frames (float64 accumulate, float32 result)."""
    n_fft, hop, win = hp["fft_size"], hp["hop_size"], hp["win_size"]
    y = wav.float().clamp(min=-1.0, max=1.0)                                            # :69
    p = int((n_fft - hop) / 2)
    y = F.pad(y.unsqueeze(1), [p, p], mode="reflect").squeeze(1)                        # :71-73
    if center:                                                                          # torch.stft(center=True, pad_mode='reflect')
        y = F.pad(y.unsqueeze(1), [n_fft // 2, n_fft // 2], mode="reflect").squeeze(1)
    window = torch.hann_window(win)
    if win < n_fft:                                                                     # torch.stft centres a short window
        left = (n_fft - win) // 2
        window = F.pad(window, [left, n_fft - win - left])
    frames = y.unfold(-1, n_fft, hop)                                                   # [B, frames, n_fft]
    spec = torch.fft.rfft(frames.double() * window.double(), dim=-1)                    # :75-76, onesided, not normalised
    re, im = spec.real.float(), spec.imag.float()
    mag = torch.sqrt(re.pow(2) + im.pow(2) + 1e-9)                                      # :81
    mel = torch.matmul(mel_basis.float(), mag.transpose(1, 2))                          # :82
    return torch.log10(torch.clamp(mel, min=1e-5))                                      # :83, :26-27
