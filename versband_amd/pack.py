"""Checkpoint -> device layout packing (host plumbing, runs once at load).

Takes the reference's ``state_dict`` tensors (SURVEY.md §8b key layout) and
produces the packed tensors the HIP kernels read: bf16 hi/lo planes for MFMA
GEMM weights, interleaved (w1,w3) rows for the fused SwiGLU epilogue, band
slices of the frequency experts, [tap][Ci][Co] conv weights, polyphase
ConvTranspose weights, folded weight-norm.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

Tensor = torch.Tensor


def to_planes(w: Tensor, n_planes: int) -> Tensor:
    """fp32 -> bf16 planes [n_planes, *w.shape]; plane 1 = bf16(w - float(plane 0))."""
    w = w.float().contiguous()
    hi = w.to(torch.bfloat16)
    if n_planes == 1:
        return hi.unsqueeze(0).contiguous()
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()


def planes_to_float(p: Tensor) -> Tensor:
    return p.float().sum(dim=0)


def pack_conv(w: Tensor) -> Tensor:
    """Conv1d weight [Co,Ci,k] -> [k][Ci][Co] fp32."""
    return w.float().permute(2, 1, 0).contiguous()


def mf_pseudo_taps(k: int) -> int:
    """pseudo-taps of a k-tap filter under F(2,3) minimal filtering: 4 per group of three taps, 2 for a lone tap, 3 for a pair."""
    return 4 * (k // 3) + (2 if k % 3 == 1 else 3 if k % 3 == 2 else 0)


def pack_conv_mf(w: Tensor) -> Tensor:
    """Conv1d weight [Co,Ci,k] -> F(2,3) minimal-filtering pseudo-taps [P][Ci][Co] fp32 (csrc/conv1d_f32w.hip, same enumeration as its
    mf_tap()): per group of three taps (w0, w1, w2): w0, (w0+w1+w2)/2, (w0-w1+w2)/2, w2; a lone last tap w: w, -w; a last pair (w0, w1):
    w0, w0+w1, -w1.  Combined in float64, rounded once."""
    co, ci, k = w.shape
    w = w.detach().double().cpu()
    taps = []
    for g in range(k // 3):
        w0, w1, w2 = w[:, :, 3 * g], w[:, :, 3 * g + 1], w[:, :, 3 * g + 2]
        taps += [w0, (w0 + w1 + w2) / 2, (w0 - w1 + w2) / 2, w2]
    o = 3 * (k // 3)
    if k % 3 == 1:
        taps += [w[:, :, o], -w[:, :, o]]
    elif k % 3 == 2:
        taps += [w[:, :, o], w[:, :, o] + w[:, :, o + 1], -w[:, :, o + 1]]
    assert len(taps) == mf_pseudo_taps(k)
    return torch.stack([t.t() for t in taps]).float().contiguous()


def pack_conv_transpose(w: Tensor, stride: int) -> Tensor:
    """ConvTranspose1d weight [Ci,Co,k] -> polyphase [stride][Kmax][Ci][Co]:
    tap j of phase p holds W[:, :, p + stride*(Kmax-1-j)] (zero when that index >= k)."""
    ci, co, k = w.shape
    kmax = (k + stride - 1) // stride
    out = torch.zeros(stride, kmax, ci, co, dtype=torch.float32, device=w.device)
    for p in range(stride):
        for j in range(kmax):
            idx = p + stride * (kmax - 1 - j)
            if idx < k:
                out[p, j] = w[:, :, idx].float()
    return out.contiguous()


def pack_conv_x3(w_packed: Tensor, ci_multiple: int = 32):
    """fp32 packed conv weights [..taps.., Ci, Co] -> split-bf16 planes [2, taps, Co, Ci_pad] (Ci contiguous, zero padded
    to a multiple of 32) for the bf16x3 MFMA conv kernel.  Returns (planes, Ci_pad)."""
    ci, co = w_packed.shape[-2], w_packed.shape[-1]
    w = w_packed.reshape(-1, ci, co).permute(0, 2, 1).float()
    cip = (ci + ci_multiple - 1) // ci_multiple * ci_multiple
    if cip != ci:
        w = torch.nn.functional.pad(w, (0, cip - ci))
    return to_planes(w.contiguous(), 2), cip


def fold_weight_norm(g: Tensor, v: Tensor) -> Tensor:
    """weight_norm(dim=0): w = g * v / ||v|| (the reference never removes weight norm:
    vocoder/hifigan/hifigan.py:15-18, so checkpoints carry weight_g / weight_v)."""
    # always on the host: the norm's reduction order differs between CPU and GPU kernels, and a rank that received its
    # checkpoint by broadcast (device tensors) must fold to the same bits as a single process that read it from disk
    dev = v.device
    g, v = g.detach().float().cpu(), v.detach().float().cpu()
    n = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return (v * (g / n)).to(dev)


def timestep_table(n: int = 1000, dim: int = 256, max_period: float = 10000.0) -> Tensor:
    """Sinusoid rows for every integer diffusion index (TimestepEmbedder.timestep_embedding,
    flag_large_dit_moe.py:110-128), computed on the CPU exactly like the reference does."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = torch.arange(n, dtype=torch.long)[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).contiguous()


def rope_tables(head_dim: int, end: int, theta: float = 10000.0):
    """cos/sin of precompute_freqs_cis (vocal2music_moe.py:436-475), CPU float32."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(end, dtype=torch.float32), freqs).float()
    return torch.cos(ang).contiguous(), torch.sin(ang).contiguous()


def pack_dit(sd: Dict[str, Tensor], cfg, n_planes: int, device) -> Dict[str, object]:
    """-> {"top": {field: tensor}, "blocks": [{field: tensor}, ...]} matching vb_dit_weights."""
    D, E, H = cfg.hidden_size, cfg.num_experts, cfg.ffn_hidden
    band = D // E
    g = lambda k: sd[k].to(device=device, dtype=torch.float32).contiguous()  # noqa: E731
    top: Dict[str, Tensor] = {}
    top["t_freq_table"] = timestep_table().to(device)
    top["t_mlp0_w"], top["t_mlp0_b"] = g("t_embedder.mlp.0.weight"), g("t_embedder.mlp.0.bias")
    top["t_mlp2_w"], top["t_mlp2_b"] = g("t_embedder.mlp.2.weight"), g("t_embedder.mlp.2.bias")
    top["adaln_w"] = torch.cat([g(f"blocks.{i}.adaLN_modulation.1.weight") for i in range(cfg.depth)]
                               + [g("final_layer.adaLN_modulation.1.weight")]).contiguous()
    top["adaln_b"] = torch.cat([g(f"blocks.{i}.adaLN_modulation.1.bias") for i in range(cfg.depth)]
                               + [g("final_layer.adaLN_modulation.1.bias")]).contiguous()
    top["adaln_wp"] = to_planes(top["adaln_w"], 2)
    top["hl_w"] = torch.cat([g(f"blocks.{i}.feed_forward.high_level_gating_network.weight") for i in range(cfg.depth)]).contiguous()
    top["hl_b"] = torch.cat([g(f"blocks.{i}.feed_forward.high_level_gating_network.bias") for i in range(cfg.depth)]).contiguous()
    top["proj_in_w"], top["proj_in_b"] = pack_conv(g("proj_in.weight")), g("proj_in.bias")
    top["proj_in_w3"] = pack_conv_x3(top["proj_in_w"])[0]
    top["final_w"], top["final_b"] = g("final_layer.linear.weight"), g("final_layer.linear.bias")
    top["final_wp"] = to_planes(top["final_w"], 2)
    cos, sin = rope_tables(cfg.head_dim, cfg.max_len)
    top["rope_cos"], top["rope_sin"] = cos.to(device), sin.to(device)
    top["midi_emb"], top["beats_emb"] = g("midi_embedding.weight"), g("beats_embedding.weight")
    top["midi_conv_w"], top["midi_conv_b"] = pack_conv(g("midi_proj.0.weight")), g("midi_proj.0.bias")
    top["beats_conv_w"], top["beats_conv_b"] = pack_conv(g("beats_proj.0.weight")), g("beats_proj.0.bias")
    top["final_proj_w"], top["final_proj_b"] = pack_conv(g("final_proj.weight")), g("final_proj.bias")
    # production (bf16) mode: the once-per-clip stem convolutions (17.7 GFLOP per clip) run on the bf16x3 kernel like every other
    # convolution of the path; the parity ("split") mode keeps them exact fp32 so the acoustic gate logits - and with them the
    # bit-identical routing of the golden vectors - do not depend on the split kernel's 3e-5
    for k in ("midi_conv_w", "beats_conv_w", "final_proj_w"):
        top[k + "3"] = pack_conv_x3(top[k])[0] if n_planes == 1 else None
    top["c_emb0"], top["c_emb0_b"] = to_planes(g("c_embedder.mlp.0.weight"), 2), g("c_embedder.mlp.0.bias")
    top["c_emb2"], top["c_emb2_b"] = to_planes(g("c_embedder.mlp.2.weight"), 2), g("c_embedder.mlp.2.bias")
    top["c_ln_w"], top["c_ln_b"] = g("c_embedder.mlp.3.weight"), g("c_embedder.mlp.3.bias")
    top["cap_ln_w"], top["cap_ln_b"] = g("cap_embedder.0.weight"), g("cap_embedder.0.bias")
    top["cap_lin_w"], top["cap_lin_b"] = g("cap_embedder.1.weight"), g("cap_embedder.1.bias")
    blocks: List[Dict[str, Tensor]] = []
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        b: Dict[str, Tensor] = {}
        b["wqkv"] = to_planes(torch.cat([g(p + "attention.wq.weight"), g(p + "attention.wk.weight"), g(p + "attention.wv.weight")]), n_planes)
        b["wo"] = to_planes(g(p + "attention.wo.weight"), n_planes)
        w_in, b_in = g(p + "feed_forward.cross_attention.in_proj_weight"), g(p + "feed_forward.cross_attention.in_proj_bias")
        b["wq_m"] = to_planes(w_in[:D], n_planes)
        b["wk_m"], b["wv_m"] = to_planes(w_in[D:2 * D], 2), to_planes(w_in[2 * D:], 2)
        qscale = float(cfg.head_dim) ** -0.5
        b["wqt_s"] = to_planes((w_in[:D].t() * qscale).contiguous(), 2)
        b["bq_s"] = (b_in[:D] * qscale).contiguous()
        b["bq_m"], b["bk_m"], b["bv_m"] = b_in[:D].contiguous(), b_in[D:2 * D].contiguous(), b_in[2 * D:].contiguous()
        b["wo_m"] = to_planes(g(p + "feed_forward.cross_attention.out_proj.weight"), n_planes)
        b["bo_m"] = g(p + "feed_forward.cross_attention.out_proj.bias")
        w13, w2 = [], []
        for grp in ("caption_experts", "acoustic_experts"):
            for e in range(E):
                q = f"{p}feed_forward.{grp}.{e}."
                w13.append(torch.stack([g(q + "w1.weight"), g(q + "w3.weight")], dim=1).reshape(2 * H, D))
                w2.append(g(q + "w2.weight"))
        b["w13"], b["w2"] = to_planes(torch.stack(w13), n_planes), to_planes(torch.stack(w2), n_planes)
        w13f, w2f = [], []
        for e in range(E):
            q = f"{p}feed_forward.freq_experts.{e}."
            lo, hi = band * e, band * (e + 1)
            w13f.append(torch.stack([g(q + "w1.weight")[:, lo:hi], g(q + "w3.weight")[:, lo:hi]], dim=1).reshape(2 * H, band))
            w2f.append(g(q + "w2.weight")[lo:hi, :])
        b["w13f"], b["w2f"] = to_planes(torch.stack(w13f), n_planes), to_planes(torch.stack(w2f), n_planes)
        b["wky"], b["wvy"] = to_planes(g(p + "attention.wk_y.weight"), 2), to_planes(g(p + "attention.wv_y.weight"), 2)
        b["attn_norm_w"], b["ffn_norm_w"] = g(p + "attention_norm.weight"), g(p + "ffn_norm.weight")
        b["y_norm_w"] = g(p + "attention_y_norm.weight")
        b["cross_w"] = torch.tanh(g(p + "attention.gate")).contiguous()
        # caption gate with MoE.cross_attention.out_proj folded in (the gate is the out_proj's only consumer):
        #   logits = (a Wo^T + bo) Wg^T + bg = a (Wg Wo)^T + (Wg bo + bg)          (vocal2music_moe.py:119-141)
        wg, bgate = g(p + "feed_forward.caption_gating_network.weight").double(), g(p + "feed_forward.caption_gating_network.bias").double()
        wo_m, bo_m = g(p + "feed_forward.cross_attention.out_proj.weight").double(), g(p + "feed_forward.cross_attention.out_proj.bias").double()
        b["wcg"], b["bcg"] = (wg @ wo_m).float().contiguous(), (wg @ bo_m + bgate).float().contiguous()
        b["wag"], b["bag"] = g(p + "feed_forward.acoustic_gating_network.weight"), g(p + "feed_forward.acoustic_gating_network.bias")
        blocks.append(b)
    return {"top": top, "blocks": blocks}


def t5_position_bias(rel_emb: Tensor, L: int, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """[heads, L, L] relative-position bias of the T5 encoder (transformers T5Attention.compute_bias with bidirectional
    buckets: half the buckets per sign, exact below num_buckets/4, logarithmic up to max_distance), built on the host."""
    ctx = torch.arange(L, dtype=torch.long)[:, None]
    mem = torch.arange(L, dtype=torch.long)[None, :]
    rel = mem - ctx
    nb = num_buckets // 2
    bucket = (rel > 0).to(torch.long) * nb
    rp = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    bucket = bucket + torch.where(rp < max_exact, rp, large)
    return rel_emb.float()[bucket].permute(2, 0, 1).contiguous()


def pack_t5(sd: Dict[str, Tensor], device, num_heads: int, max_len: int = 128, num_buckets: int = 32, max_distance: int = 128):
    """transformers.T5EncoderModel state_dict -> tensors of vb_t5_weights (all GEMM operands as 2 bf16 planes)."""
    g = lambda k: sd[k].to(device=device, dtype=torch.float32).contiguous()      # noqa: E731
    n_layers = len({int(k.split(".")[2]) for k in sd if k.startswith("encoder.block.")})
    top = {"embed": g("shared.weight") if "shared.weight" in sd else g("encoder.embed_tokens.weight"),
           "pos_bias": t5_position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].cpu(), max_len,
                                        num_buckets, max_distance).to(device),
           "final_ln": g("encoder.final_layer_norm.weight")}
    top["ones"] = torch.ones(top["embed"].shape[1], dtype=torch.float32, device=device)
    layers = []
    for i in range(n_layers):
        p = f"encoder.block.{i}."
        wi0, wi1 = g(p + "layer.1.DenseReluDense.wi_0.weight"), g(p + "layer.1.DenseReluDense.wi_1.weight")
        layers.append({
            "ln0": g(p + "layer.0.layer_norm.weight"),
            "wqkv": to_planes(torch.cat([g(p + f"layer.0.SelfAttention.{n}.weight") for n in ("q", "k", "v")]), 2),
            "wo": to_planes(g(p + "layer.0.SelfAttention.o.weight"), 2),
            "ln1": g(p + "layer.1.layer_norm.weight"),
            "wi": to_planes(torch.stack([wi0, wi1], dim=1).reshape(2 * wi0.shape[0], wi0.shape[1]).contiguous(), 2),
            "wo_ff": to_planes(g(p + "layer.1.DenseReluDense.wo.weight"), 2),
        })
    return top, layers


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> Tensor:
    """vocoder/bigvgan/alias_free_torch/filter.py:28-57: the Kaiser-windowed sinc low-pass of BigVGAN's anti-aliased activations."""
    half_size = kernel_size // 2
    A = 2.285 * (half_size - 1) * math.pi * (4 * half_width) + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if kernel_size % 2 == 0 else torch.arange(kernel_size) - half_size
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (f / f.sum()).float()
