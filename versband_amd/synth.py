"""Synthetic checkpoints and inputs (seeded by the portable PRNG).

There is no network for real checkpoints, so parity tests, the benchmark and
the golden fixtures all use weights generated here.  Key names and shapes are
the reference's ``state_dict`` layout:

* DiT      : ``TxtFlagLargeImprovedDiTV2``  ldm/modules/diffusionmodules/vocal2music_moe.py:293-374
* VAE dec  : ``AutoencoderKL`` decode side   ldm/models/autoencoder1d.py:14-58,411-478
* HiFi-GAN : ``HifiGanGenerator`` (weight-norm g/v pairs) vocoder/hifigan/modules/hifigan.py:101-124

Zero-initialised parameters of the reference (adaLN, final layer, attention
gate - SURVEY Q8) get NON-zero values here, otherwise every block is the
identity and parity would be trivial.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import prng

# ----------------------------------------------------------------------------
# configs
# ----------------------------------------------------------------------------


@dataclass
class DiTConfig:
    """params of configs/vocal2music.yaml:36-43 plus fixed constructor defaults."""
    in_channels: int = 20
    ori_dim: int = 1024          # T5 width; not forwarded by the reference ctor
    context_dim: int = 768
    hidden_size: int = 768
    num_heads: int = 8
    depth: int = 4
    max_len: int = 1500
    num_experts: int = 4
    multiple_of: int = 256
    norm_eps: float = 1e-5

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def ffn_hidden(self) -> int:
        # FeedForward.__init__ flag_large_dit_moe.py:462-468 with hidden_dim = dim
        h = int(2 * self.hidden_size / 3)
        return self.multiple_of * ((h + self.multiple_of - 1) // self.multiple_of)


@dataclass
class VAEConfig:
    """ddconfig of configs/vocal2music.yaml:51-66."""
    embed_dim: int = 20
    z_channels: int = 20
    in_channels: int = 80
    out_ch: int = 80
    kernel_size: int = 5
    ch: int = 384
    ch_mult: Tuple[int, ...] = (1, 2, 4)
    num_res_blocks: int = 2
    attn_layers: Tuple[int, ...] = (3,)
    down_layers: Tuple[int, ...] = (0,)


@dataclass
class T5Config:
    """google/flan-t5-large encoder (cond_stage_config of configs/vocal2music.yaml:71-74; HF T5Config of that checkpoint)."""
    vocab_size: int = 32128
    d_model: int = 1024
    d_kv: int = 64
    num_heads: int = 16
    d_ff: int = 2816
    num_layers: int = 24
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6


def t5_encoder_shapes(cfg: "T5Config"):
    """state_dict keys of transformers.T5EncoderModel (shared.weight is tied to encoder.embed_tokens.weight)."""
    s: "OrderedDict[str, Tuple[Tuple[int, ...], tuple]]" = OrderedDict()
    inner = cfg.num_heads * cfg.d_kv
    s["shared.weight"] = ((cfg.vocab_size, cfg.d_model), ("u", 1.0))
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}."
        for nm in ("q", "k", "v"):
            s[p + f"layer.0.SelfAttention.{nm}.weight"] = ((inner, cfg.d_model), ("fan",))
        s[p + "layer.0.SelfAttention.o.weight"] = ((cfg.d_model, inner), ("fan",))
        if i == 0:
            s[p + "layer.0.SelfAttention.relative_attention_bias.weight"] = ((cfg.relative_attention_num_buckets, cfg.num_heads), ("u", 1.0))
        s[p + "layer.0.layer_norm.weight"] = ((cfg.d_model,), ("norm",))
        s[p + "layer.1.DenseReluDense.wi_0.weight"] = ((cfg.d_ff, cfg.d_model), ("fan",))
        s[p + "layer.1.DenseReluDense.wi_1.weight"] = ((cfg.d_ff, cfg.d_model), ("fan",))
        s[p + "layer.1.DenseReluDense.wo.weight"] = ((cfg.d_model, cfg.d_ff), ("fan",))
        s[p + "layer.1.layer_norm.weight"] = ((cfg.d_model,), ("norm",))
    s["encoder.final_layer_norm.weight"] = ((cfg.d_model,), ("norm",))
    return s


@dataclass
class HifiGanConfig:
    """<vocoder_ckpt>/config.yaml keys read by HifiGanGenerator.
    Defaults = the synthetic V1-like config of SURVEY §8(d)."""
    resblock: str = "1"
    upsample_rates: Tuple[int, ...] = (8, 5, 4, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 15, 8, 4)
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    num_mels: int = 80

    def as_hparams(self) -> dict:
        return {
            "resblock": self.resblock,
            "upsample_rates": list(self.upsample_rates),
            "upsample_kernel_sizes": list(self.upsample_kernel_sizes),
            "upsample_initial_channel": self.upsample_initial_channel,
            "resblock_kernel_sizes": list(self.resblock_kernel_sizes),
            "resblock_dilation_sizes": [list(d) for d in self.resblock_dilation_sizes],
        }

    @property
    def hop(self) -> int:
        h = 1
        for u in self.upsample_rates:
            h *= u
        return h


# init kinds: ("u", a) uniform(-a,a); ("norm",) 1+U(-.2,.2); ("fan",) U(+-1/sqrt(fan_in));
#             ("pos",) U(0.5,1.5)


def dit_shapes(cfg: DiTConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], tuple]]":
    D, H, E = cfg.hidden_size, cfg.ffn_hidden, cfg.num_experts
    a = 0.02 * 3 ** 0.5
    s: "OrderedDict[str, Tuple[Tuple[int, ...], tuple]]" = OrderedDict()

    def lin(name, o, i, bias=True, scale=a):
        s[name + ".weight"] = ((o, i), ("u", scale))
        if bias:
            s[name + ".bias"] = ((o,), ("u", 0.02))

    lin("t_embedder.mlp.0", D, 256)
    lin("t_embedder.mlp.2", D, D)
    s["proj_in.weight"] = ((D, cfg.in_channels, 5), ("u", a * 3))
    s["proj_in.bias"] = ((D,), ("u", 0.02))
    s["code_proj.0.weight"] = ((D, cfg.in_channels, 5), ("u", a))   # dead weight
    s["code_proj.0.bias"] = ((D,), ("u", 0.02))
    s["midi_embedding.weight"] = ((130, D), ("u", 1.0))
    s["beats_embedding.weight"] = ((3, D), ("u", 1.0))
    for nm in ("midi_proj", "beats_proj"):
        s[nm + ".0.weight"] = ((D, D, 5), ("u", a * 0.5))
        s[nm + ".0.bias"] = ((D,), ("u", 0.02))
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        for w in ("wq", "wk", "wv"):
            lin(p + "attention." + w, D, D, bias=False, scale=a * 1.5)
        lin(p + "attention.wk_y", D, cfg.context_dim, bias=False, scale=a * 1.5)
        lin(p + "attention.wv_y", D, cfg.context_dim, bias=False)
        s[p + "attention.gate"] = ((cfg.num_heads,), ("u", 1.0))
        lin(p + "attention.wo", D, D, bias=False)
        lin(p + "feed_forward.high_level_gating_network", 2, D)
        for grp in ("caption_experts", "acoustic_experts"):
            for e in range(E):
                q = f"{p}feed_forward.{grp}.{e}."
                lin(q + "w1", H, D, bias=False)
                lin(q + "w2", D, H, bias=False)
                lin(q + "w3", H, D, bias=False)
        lin(p + "feed_forward.caption_gating_network", E, D, scale=a * 2)
        lin(p + "feed_forward.acoustic_gating_network", E, D, scale=a * 2)
        for e in range(E):
            q = f"{p}feed_forward.freq_experts.{e}."
            lin(q + "w1", H, D, bias=False, scale=a * 2)
            lin(q + "w2", D, H, bias=False)
            lin(q + "w3", H, D, bias=False, scale=a * 2)
        s[p + "feed_forward.cross_attention.in_proj_weight"] = ((3 * D, D), ("u", a * 1.5))
        s[p + "feed_forward.cross_attention.in_proj_bias"] = ((3 * D,), ("u", 0.02))
        lin(p + "feed_forward.cross_attention.out_proj", D, D)
        s[p + "attention_norm.weight"] = ((D,), ("norm",))
        s[p + "ffn_norm.weight"] = ((D,), ("norm",))
        lin(p + "adaLN_modulation.1", 6 * D, D, scale=a * 0.5)
        s[p + "attention_y_norm.weight"] = ((cfg.context_dim,), ("norm",))
    s["final_proj.weight"] = ((D, D, 1), ("u", a))
    s["final_proj.bias"] = ((D,), ("u", 0.02))
    lin("final_layer.linear", cfg.in_channels, D)
    lin("final_layer.adaLN_modulation.1", 2 * D, D, scale=a * 0.5)
    s["cap_embedder.0.weight"] = ((cfg.context_dim,), ("norm",))
    s["cap_embedder.0.bias"] = ((cfg.context_dim,), ("u", 0.02))
    lin("cap_embedder.1", D, cfg.context_dim)
    lin("c_embedder.mlp.0", D, cfg.ori_dim)
    lin("c_embedder.mlp.2", D, D)
    s["c_embedder.mlp.3.weight"] = ((D,), ("norm",))
    s["c_embedder.mlp.3.bias"] = ((D,), ("u", 0.02))
    return s


def vae_decoder_shapes(cfg: VAEConfig):
    """post_quant_conv + decoder.* keys of AutoencoderKL (autoencoder1d.py:31,411-478)."""
    s: "OrderedDict[str, Tuple[Tuple[int, ...], tuple]]" = OrderedDict()

    def conv(name, co, ci, k):
        s[name + ".weight"] = ((co, ci, k), ("fan",))
        s[name + ".bias"] = ((co,), ("u", 0.02))

    def norm(name, c):
        s[name + ".weight"] = ((c,), ("norm",))
        s[name + ".bias"] = ((c,), ("u", 0.05))

    def res(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", co, ci, 3)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co, 3)
        if ci != co:
            conv(name + ".nin_shortcut", co, ci, 1)

    conv("post_quant_conv", cfg.z_channels, cfg.embed_dim, 1)
    nl = len(cfg.ch_mult)
    block_in = cfg.ch * cfg.ch_mult[nl - 1]
    conv("decoder.conv_in", block_in, cfg.z_channels, cfg.kernel_size)
    res("decoder.mid.block_1", block_in, block_in)
    norm("decoder.mid.attn_1.norm", block_in)
    for nm in ("q", "k", "v", "proj_out"):
        conv("decoder.mid.attn_1." + nm, block_in, block_in, 1)
    res("decoder.mid.block_2", block_in, block_in)
    up_levels = [d + 1 for d in cfg.down_layers]
    for lvl in reversed(range(nl)):
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks + 1):
            res(f"decoder.up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if lvl in cfg.attn_layers:
                norm(f"decoder.up.{lvl}.attn.{b}.norm", block_in)
                for nm in ("q", "k", "v", "proj_out"):
                    conv(f"decoder.up.{lvl}.attn.{b}." + nm, block_in, block_in, 1)
        if lvl in up_levels:
            conv(f"decoder.up.{lvl}.upsample.conv", block_in, block_in, 3)
    norm("decoder.norm_out", block_in)
    conv("decoder.conv_out", cfg.out_ch, block_in, cfg.kernel_size)
    return s



def vae_encoder_shapes(cfg: VAEConfig):
    """encoder.* + quant_conv keys of AutoencoderKL (autoencoder1d.py:30,315-381).  Encoder ResnetBlocks take the
    ddconfig kernel_size (:346-351), unlike the decoder's default 3."""
    s: "OrderedDict[str, Tuple[Tuple[int, ...], tuple]]" = OrderedDict()

    def conv(name, co, ci, k):
        s[name + ".weight"] = ((co, ci, k), ("fan",))
        s[name + ".bias"] = ((co,), ("u", 0.02))

    def norm(name, c):
        s[name + ".weight"] = ((c,), ("norm",))
        s[name + ".bias"] = ((c,), ("u", 0.05))

    def res(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", co, ci, cfg.kernel_size)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co, cfg.kernel_size)
        if ci != co:
            conv(name + ".nin_shortcut", co, ci, 1)

    def attn(name, c):
        norm(name + ".norm", c)
        for nm in ("q", "k", "v", "proj_out"):
            conv(name + "." + nm, c, c, 1)

    conv("encoder.conv_in", cfg.ch, cfg.in_channels, cfg.kernel_size)
    in_mult = (1,) + tuple(cfg.ch_mult)
    block_in = cfg.ch
    for lvl in range(len(cfg.ch_mult)):
        block_in = cfg.ch * in_mult[lvl]
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks):
            res(f"encoder.down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if lvl in cfg.attn_layers:
                attn(f"encoder.down.{lvl}.attn.{b}", block_in)
        if lvl in cfg.down_layers:
            conv(f"encoder.down.{lvl}.downsample.conv", block_in, block_in, 3)
    res("encoder.mid.block_1", block_in, block_in)
    attn("encoder.mid.attn_1", block_in)
    res("encoder.mid.block_2", block_in, block_in)
    norm("encoder.norm_out", block_in)
    conv("encoder.conv_out", 2 * cfg.z_channels, block_in, cfg.kernel_size)
    conv("quant_conv", 2 * cfg.embed_dim, 2 * cfg.z_channels, 1)
    return s

def hifigan_shapes(cfg: HifiGanConfig):
    s: "OrderedDict[str, Tuple[Tuple[int, ...], tuple]]" = OrderedDict()

    def wn(name, shape, fan_in):
        s[name + ".bias"] = ((shape[0] if "ups." not in name else shape[1],), ("u", 0.02))
        s[name + ".weight_g"] = ((shape[0], 1, 1), ("pos",))
        s[name + ".weight_v"] = (shape, ("u", 1.0 / fan_in ** 0.5))

    c0 = cfg.upsample_initial_channel
    wn("conv_pre", (c0, cfg.num_mels, 7), cfg.num_mels * 7)
    ch = c0
    nk = len(cfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        wn(f"ups.{i}", (cin, cout, k), cin * k / u)
        ch = cout
        for j, (rk, rd) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            n = i * nk + j
            if cfg.resblock == "1":
                for m in range(len(rd)):
                    wn(f"resblocks.{n}.convs1.{m}", (ch, ch, rk), ch * rk)
                for m in range(len(rd)):
                    wn(f"resblocks.{n}.convs2.{m}", (ch, ch, rk), ch * rk)
            else:
                for m in range(len(rd)):
                    wn(f"resblocks.{n}.convs.{m}", (ch, ch, rk), ch * rk)
    wn("conv_post", (1, ch, 7), ch * 7)
    return s


@dataclass
class BigVGANConfig:
    """args.yml keys read by vocoder/bigvgan/models.py BigVGAN (synthetic config: same geometry as the HiFi-GAN one, hop 320)."""
    resblock: str = "1"
    upsample_rates: Tuple[int, ...] = (8, 5, 4, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 15, 8, 4)
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    num_mels: int = 80
    activation: str = "snakebeta"
    snake_logscale: bool = True

    def as_hparams(self) -> dict:
        return {"resblock": self.resblock, "upsample_rates": list(self.upsample_rates), "upsample_kernel_sizes": list(self.upsample_kernel_sizes),
                "upsample_initial_channel": self.upsample_initial_channel, "resblock_kernel_sizes": list(self.resblock_kernel_sizes),
                "resblock_dilation_sizes": [list(d) for d in self.resblock_dilation_sizes], "num_mels": self.num_mels,
                "activation": self.activation, "snake_logscale": self.snake_logscale}


def bigvgan_shapes(cfg: BigVGANConfig):
    """Trainable keys of BigVGAN.state_dict() (vocoder/bigvgan/models.py:133-170); the anti-aliasing filters are buffers the
    module computes itself and are not listed."""
    s: "OrderedDict[str, Tuple[Tuple[int, ...], tuple]]" = OrderedDict()

    def wn(name, shape, fan_in, transposed=False):
        s[name + ".bias"] = ((shape[1] if transposed else shape[0],), ("u", 0.02))
        s[name + ".weight_g"] = ((shape[0], 1, 1), ("pos",))
        s[name + ".weight_v"] = (shape, ("u", 1.0 / fan_in ** 0.5))

    def act(name, ch):
        s[name + ".act.alpha"] = ((ch,), ("u", 0.7))                 # log-scale (or linear around 1 +- handled by the kind below)
        if cfg.activation == "snakebeta":
            s[name + ".act.beta"] = ((ch,), ("u", 0.7))

    c0 = cfg.upsample_initial_channel
    wn("conv_pre", (c0, cfg.num_mels, 7), cfg.num_mels * 7)
    ch = c0
    nk = len(cfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        wn(f"ups.{i}.0", (cin, cout, k), cin * k / u, transposed=True)
        ch = cout
        for j, (rk, rd) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            n = i * nk + j
            if cfg.resblock == "1":
                for m in range(len(rd)):
                    wn(f"resblocks.{n}.convs1.{m}", (ch, ch, rk), ch * rk)
                for m in range(len(rd)):
                    wn(f"resblocks.{n}.convs2.{m}", (ch, ch, rk), ch * rk)
                for m in range(2 * len(rd)):
                    act(f"resblocks.{n}.activations.{m}", ch)
            else:
                for m in range(len(rd)):
                    wn(f"resblocks.{n}.convs.{m}", (ch, ch, rk), ch * rk)
                for m in range(len(rd)):
                    act(f"resblocks.{n}.activations.{m}", ch)
    act("activation_post", ch)
    wn("conv_post", (1, ch, 7), ch * 7)
    s["conv_post.weight_g"] = ((1, 1, 1), ("pos", 0.07))      # keeps the synthetic generator's tanh input O(0.5): unsaturated output
    return s


def make_state_dict(shapes, seed: int, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, (shape, kind) in shapes.items():
        n = int(np.prod(shape))
        ks = prng.key_seed(seed, name)
        if kind[0] == "u":
            v = prng.uniform(ks, n, -kind[1], kind[1])
        elif kind[0] == "norm":
            v = prng.uniform(ks, n, 0.8, 1.2)
        elif kind[0] == "pos":
            v = prng.uniform(ks, n, 0.5, 1.5) * (kind[1] if len(kind) > 1 else 1.0)
        elif kind[0] == "fan":
            fan_in = int(np.prod(shape[1:]))
            b = 1.0 / fan_in ** 0.5
            v = prng.uniform(ks, n, -b, b)
        else:
            raise ValueError(kind)
        sd[prefix + name] = torch.from_numpy(v.reshape(shape).copy())
    return sd


# ----------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------


def make_clip_inputs(seed: int, clip: int, T_lat: int, L: int = 80, ori_dim: int = 1024, C: int = 20,
                     valid_mel: int | None = None) -> Dict[str, torch.Tensor]:
    """One clip's inputs keyed by the GLOBAL clip index (world-size independent).

    x_latent [C,T], t5_cond/t5_uncond [L,ori_dim], midi/beats [1,2T] int64.
    midi held in runs of 15..60 frames over 0..128, beats has a 1 every ~37
    frames; frames >= valid_mel carry the training pad values 128 / 2.
    """
    s = seed + clip
    T_mel = 2 * T_lat
    if valid_mel is None:
        valid_mel = T_mel - 4 if T_mel >= 8 else T_mel
    x = prng.normal(prng.key_seed(s, "x_latent"), C * T_lat).reshape(C, T_lat)
    t5c = prng.normal(prng.key_seed(s, "t5_cond"), L * ori_dim).reshape(L, ori_dim)
    t5u = prng.normal(prng.key_seed(s, "t5_uncond"), L * ori_dim).reshape(L, ori_dim)
    runs = prng.randint(prng.key_seed(s, "midi_runs"), T_mel, 15, 61)
    vals = prng.randint(prng.key_seed(s, "midi_vals"), T_mel, 0, 129)
    midi = np.empty(T_mel, dtype=np.int64)
    pos, r = 0, 0
    while pos < T_mel:
        n = int(runs[r])
        midi[pos:pos + n] = vals[r]
        pos += n
        r += 1
    beats = np.zeros(T_mel, dtype=np.int64)
    jit = prng.randint(prng.key_seed(s, "beat_jit"), T_mel // 30 + 2, -2, 3)
    for i in range(T_mel // 37 + 1):
        p = i * 37 + int(jit[i]) + 2
        if 0 <= p < T_mel:
            beats[p] = 1
    midi[valid_mel:] = 128
    beats[valid_mel:] = 2
    return {
        "x_latent": torch.from_numpy(x.copy()),
        "t5_cond": torch.from_numpy(t5c.copy()),
        "t5_uncond": torch.from_numpy(t5u.copy()),
        "midi": torch.from_numpy(midi.reshape(1, T_mel)),
        "beats": torch.from_numpy(beats.reshape(1, T_mel)),
    }


def gumbel_exponentials(seed: int, clip: int, nfe: int, block: int, gate: int, T: int, width: int) -> np.ndarray:
    """Exp(1) draws for one (clip, nfe, block, gate): [T, width] float32.

    gate 0 = high-level [T,2], 1 = caption [T,E], 2 = acoustic [T,E]
    (draw order of MoE.forward, vocal2music_moe.py:134,150,151)."""
    ks = prng.key_seed(seed + clip, f"gumbel/{nfe}/{block}/{gate}")
    return prng.exponential(ks, T * width).reshape(T, width)
