#!/usr/bin/env python3
"""Generate golden fixtures by importing the REAL reference (build container only).

Runs only where /root/reference exists; never on the GPU box.  The reference's
source is never copied: this script imports it with inert stubs for the missing
third-party modules, loads the synthetic weights of
``versband_amd.synth`` into the reference modules, injects the PRNG noise by
patching ``Tensor.exponential_`` and stores inputs + reference outputs as small
``tests/golden/*.npz`` files (data only).

    python oracle/gen_golden.py            # writes tests/golden/*.npz
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from collections import deque

os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from versband_amd import prng, synth  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


# ---------------------------------------------------------------------------
# stubs
# ---------------------------------------------------------------------------

def _mod(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__vb_stub__ = True
    # a real module spec: importlib.util.find_spec() (transformers probes optional packages with it) raises on __spec__ = None
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Euler:
    """Fixed-step explicit Euler standing in for torchdyn.NeuralODE (absent here and
    un-pinned upstream - SURVEY §8 A4): x <- x + dt f(t, x), left endpoints only."""

    def __init__(self, vf, solver="euler", sensitivity=None, atol=None, rtol=None):
        assert solver == "euler"
        self.vf = vf

    def __call__(self, x, t_span):
        sol = [x]
        t = t_span[0]
        for k in range(len(t_span) - 1):
            dt = t_span[k + 1] - t
            x = x + dt * self.vf(t, x, args={})
            t = t + dt
            sol.append(x)
        return t_span, torch.stack(sol)


def install_stubs():
    import transformers  # noqa: F401  (must be imported before torchvision is stubbed)
    # resolve transformers' lazily imported T5 classes NOW, before flash_attn / torchvision are replaced by stubs: their import
    # chain probes those packages, so gen_t5() after install_stubs() used to fail unless it was run on its own
    from transformers import T5Config, T5EncoderModel  # noqa: F401
    _mod("flash_attn", flash_attn_func=None, flash_attn_varlen_func=None)
    _mod("flash_attn.bert_padding", index_first_axis=None, pad_input=None, unpad_input=None)

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, Callback=object, Trainer=object,
              seed_everything=lambda *a, **k: None)
    ident = lambda f: f  # noqa: E731
    _mod("pytorch_lightning.utilities", rank_zero_only=ident)
    _mod("pytorch_lightning.utilities.distributed", rank_zero_only=ident)
    _mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=ident, rank_zero_info=print)
    pl.utilities = sys.modules["pytorch_lightning.utilities"]

    class _Prof:
        def __init__(self, *a, **k):
            pass
    _mod("pytorch_memlab", LineProfiler=_Prof, profile=ident)
    tv = _mod("torchvision")
    tvu = _mod("torchvision.utils", make_grid=lambda *a, **k: None)
    tv.utils = tvu
    _mod("taming")
    _mod("taming.modules")
    _mod("taming.modules.vqvae")
    _mod("taming.modules.vqvae.quantize", VectorQuantizer2=object, VectorQuantizer=object, GumbelQuantize=object)
    _mod("icecream", ic=lambda *a, **k: None)
    _mod("omegaconf", ListConfig=list, OmegaConf=object, DictConfig=dict)
    _mod("importlib_resources", files=lambda *a, **k: None)
    _mod("torchdyn")
    _mod("torchdyn.core", NeuralODE=_Euler)

    class FrozenTextVocalEmbedder(nn.Module):
        """dummy T5: passes pre-computed embeddings through (BASELINE: dummy T5 emb)."""

        def __init__(self, *a, **k):
            super().__init__()
            self.device = "cpu"

        def forward(self, c):
            return c

        def encode(self, c):
            return c
    _mod("ldm.modules.encoders.modules", FrozenTextVocalEmbedder=FrozenTextVocalEmbedder)
    torch.Tensor.cuda = lambda self, *a, **k: self


class NoiseQueue:
    """Feeds Tensor.exponential_ from a queue of pre-drawn arrays (FIFO)."""

    def __init__(self):
        self.q = deque()
        self._orig = torch.Tensor.exponential_
        q = self.q

        def patched(t, *a, **k):
            src = q.popleft()
            assert tuple(src.shape) == tuple(t.shape), (src.shape, t.shape)
            t.copy_(src)
            return t
        torch.Tensor.exponential_ = patched

    def push(self, arr):
        self.q.append(torch.as_tensor(arr))

    def restore(self):
        torch.Tensor.exponential_ = self._orig


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


# ---------------------------------------------------------------------------
# fixture builders
# ---------------------------------------------------------------------------

SEED = 1234


def block_noise(B, T, E, nfe, depth, clip0=0):
    """per block (E1[N,2], E2[N,E], E3[N,E]); rows ordered b-major like x.reshape(-1,D)."""
    out = []
    for i in range(depth):
        parts = []
        for gate, w in ((0, 2), (1, E), (2, E)):
            parts.append(np.concatenate([synth.gumbel_exponentials(SEED, clip0 + b, nfe, i, gate, T, w) for b in range(B)], 0))
        out.append(tuple(parts))
    return out


def clip_batch(B, T, L, clip0=0):
    clips = [synth.make_clip_inputs(SEED, clip0 + b, T, L=L) for b in range(B)]
    st = lambda k: torch.stack([c[k] for c in clips])  # noqa: E731
    return {k: st(k) for k in clips[0]}


def gen_dit(E: int, tag: str, B=2, T=24, L=8):
    from ldm.modules.diffusionmodules.vocal2music_moe import TxtFlagLargeImprovedDiTV2
    cfg = synth.DiTConfig(num_experts=E)
    net = TxtFlagLargeImprovedDiTV2(in_channels=cfg.in_channels, context_dim=cfg.context_dim, hidden_size=cfg.hidden_size,
                                    depth=cfg.depth, num_heads=cfg.num_heads, max_len=cfg.max_len, num_experts=E,
                                    ori_dim=cfg.ori_dim).eval()
    sd = synth.make_state_dict(synth.dit_shapes(cfg), SEED)
    ref_sd = net.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()), (set(ref_sd) ^ set(sd))
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd, strict=True)
    inp = clip_batch(B, T, L)
    nq = NoiseQueue()
    outs = {}
    t_idx = torch.tensor([416] * B, dtype=torch.long)
    for br, t5 in ((0, inp["t5_cond"]), (1, inp["t5_uncond"])):
        noise = block_noise(B, T, E, nfe=br, depth=cfg.depth)
        for blk in noise:
            for a in blk:
                nq.push(a)
        ctx = {"c_concat": {"midi": inp["midi"], "beats": inp["beats"]}, "c_crossattn": t5, "name": ["x"] * B}
        with torch.no_grad():
            v, _ = net(inp["x_latent"], t_idx, ctx)
        outs[f"v{br}"] = v.numpy()
        for i, blk in enumerate(noise):
            for j, a in enumerate(blk):
                outs[f"noise{br}_{i}_{j}"] = a
    assert len(nq.q) == 0
    nq.restore()
    # routing indices & intermediate activations via hooks would need reference edits;
    # instead record them from a second pass with forward hooks on the MoE modules
    rec = {}

    def mk(i):
        def hook(mod, args, out):
            rec[f"moe_out{i}"] = out[0].detach().numpy().copy()
        return hook
    hs = [blk.feed_forward.register_forward_hook(mk(i)) for i, blk in enumerate(net.blocks)]
    nq = NoiseQueue()
    for blk in block_noise(B, T, E, nfe=0, depth=cfg.depth):
        for a in blk:
            nq.push(a)
    ctx = {"c_concat": {"midi": inp["midi"], "beats": inp["beats"]}, "c_crossattn": inp["t5_cond"], "name": ["x"] * B}
    with torch.no_grad():
        net(inp["x_latent"], t_idx, ctx)
    nq.restore()
    for h in hs:
        h.remove()
    outs.update(rec)
    outs.update({"x": inp["x_latent"].numpy(), "t5_cond": inp["t5_cond"].numpy(), "t5_uncond": inp["t5_uncond"].numpy(),
                 "midi": inp["midi"].numpy(), "beats": inp["beats"].numpy(), "t_idx": t_idx.numpy(),
                 "meta": np.array([B, T, L, E, SEED], dtype=np.int64)})
    np.savez_compressed(os.path.join(GOLD, f"dit_forward_{tag}.npz"), **outs)
    print("dit", tag, {k: v.shape for k, v in outs.items() if k.startswith("v")}, float(np.abs(outs["v0"]).max()))
    return net, sd, cfg


def gen_sampler(B=2, T=16, L=8, steps=3, scale=3.0):
    """Full CFM path: instantiate_from_config(configs/vocal2music.yaml model) ->
    CFMSampler.sample_cfg -> decode_first_stage, with the reference's own classes."""
    import yaml
    from ldm.util import instantiate_from_config
    from ldm.models.diffusion.cfm1_audio_sampler import CFMSampler
    with open(os.path.join(REF, "configs", "vocal2music.yaml")) as f:
        conf = yaml.safe_load(f)["model"]
    conf["params"]["first_stage_config"]["params"]["ckpt_path"] = None
    model = instantiate_from_config(conf).eval()
    dcfg, vcfg = synth.DiTConfig(), synth.VAEConfig()
    sd_dit = synth.make_state_dict(synth.dit_shapes(dcfg), SEED)
    sd_vae = synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1)
    full = {}
    full.update({"model.diffusion_model." + k: v for k, v in sd_dit.items()})
    full.update({"first_stage_model." + k: v for k, v in sd_vae.items()})
    full["scale_factor"] = torch.tensor(0.75)
    missing, unexpected = model.load_state_dict(full, strict=False)
    assert not unexpected, unexpected
    assert not [m for m in missing if m.startswith("model.diffusion_model.") or m.startswith("first_stage_model.decoder")
                or m.startswith("first_stage_model.post_quant")], missing
    sampler = CFMSampler(model, num_timesteps=1000)
    inp = clip_batch(B, T, L)
    E = dcfg.num_experts
    nq = NoiseQueue()
    for k in range(steps):
        for br in (0, 1):
            for blk in block_noise(B, T, E, nfe=2 * k + br, depth=dcfg.depth):
                for a in blk:
                    nq.push(a)
    ac = {"acoustic": torch.zeros(B, 20, 2 * T), "midi": inp["midi"], "beats": inp["beats"]}
    c = model.get_learned_conditioning({"caption": inp["t5_cond"], "acoustic": ac, "name": ["x"] * B})
    uc = model.get_learned_conditioning({"caption": inp["t5_uncond"], "acoustic": ac, "name": ["x"] * B})
    z, traj = sampler.sample_cfg(S=100, cond=c, batch_size=B, shape=[20, T], verbose=False,
                                 unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                 x_T=inp["x_latent"], x_latent=inp["x_latent"], timesteps=steps + 1)
    assert len(nq.q) == 0
    nq.restore()
    with torch.no_grad():
        mel = model.decode_first_stage(z)
    outs = {"z": z.numpy(), "traj": traj.numpy(), "mel": mel.numpy(), "x": inp["x_latent"].numpy(),
            "t5_cond": inp["t5_cond"].numpy(), "t5_uncond": inp["t5_uncond"].numpy(), "midi": inp["midi"].numpy(),
            "beats": inp["beats"].numpy(),
            "meta": np.array([B, T, L, E, SEED, steps], dtype=np.int64), "scale": np.float32(scale),
            "scale_factor": np.float32(0.75)}
    np.savez_compressed(os.path.join(GOLD, "sample_cfg_3step.npz"), **outs)
    print("sampler z", z.shape, float(z.abs().max()), "mel", mel.shape, float(mel.abs().max()))
    # t-index tables through the reference's own Wrapper arithmetic (cfm1_audio.py:156)
    tabs = {}
    for n in (10, 24, 50):
        ts = torch.linspace(0, 1, n + 1)
        tabs[f"tidx{n}"] = np.array([int(torch.tensor([ts[k] * 1000] * 1).long()[0]) for k in range(n)], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "t_index_tables.npz"), **tabs)
    return model


def gen_vae(B=2, T=16):
    from ldm.models.autoencoder1d import AutoencoderKL
    vcfg = synth.VAEConfig()
    dd = dict(double_z=True, in_channels=80, out_ch=80, z_channels=20, kernel_size=5, ch=384, ch_mult=[1, 2, 4],
              num_res_blocks=2, attn_layers=[3], down_layers=[0], dropout=0.0)
    ae = AutoencoderKL(embed_dim=20, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}).eval()
    sd = synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1)
    ref = {k: v for k, v in ae.state_dict().items() if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    assert set(ref) == set(sd), set(ref) ^ set(sd)
    ae.load_state_dict(sd, strict=False)
    z = torch.from_numpy(prng.normal(prng.key_seed(SEED, "vae_z"), B * 20 * T).reshape(B, 20, T))
    with torch.no_grad():
        mel = ae.decode(z)
    np.savez_compressed(os.path.join(GOLD, "vae_decode.npz"), z=z.numpy(), mel=mel.numpy())
    print("vae", mel.shape, float(mel.abs().max()))


def gen_vae_encode(B=2, T_mel=32):
    from ldm.models.autoencoder1d import AutoencoderKL
    vcfg = synth.VAEConfig()
    dd = dict(double_z=True, in_channels=80, out_ch=80, z_channels=20, kernel_size=5, ch=384, ch_mult=[1, 2, 4],
              num_res_blocks=2, attn_layers=[3], down_layers=[0], dropout=0.0)
    ae = AutoencoderKL(embed_dim=20, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}).eval()
    sd = synth.make_state_dict(synth.vae_encoder_shapes(vcfg), SEED + 3)
    ref = {k: v for k, v in ae.state_dict().items() if k.startswith("encoder.") or k.startswith("quant_conv.")}
    assert set(ref) == set(sd), set(ref) ^ set(sd)
    for k in ref:
        assert tuple(ref[k].shape) == tuple(sd[k].shape), (k, ref[k].shape, sd[k].shape)
    ae.load_state_dict(sd, strict=False)
    x = torch.from_numpy(prng.normal(prng.key_seed(SEED, "vae_enc_x"), B * 80 * T_mel).reshape(B, 80, T_mel))
    eps = torch.from_numpy(prng.normal(prng.key_seed(SEED, "vae_enc_eps"), B * 20 * (T_mel // 2)).reshape(B, 20, T_mel // 2))
    with torch.no_grad():
        post = ae.encode(x)
        z = post.mean + post.std * eps            # DiagonalGaussianDistribution.sample with the noise injected
    np.savez_compressed(os.path.join(GOLD, "vae_encode.npz"), x=x.numpy(), moments=post.parameters.numpy(), eps=eps.numpy(),
                        z=z.numpy(), mode=post.mode().numpy())
    print("vae_encode", post.parameters.shape, float(post.parameters.abs().max()))


def gen_hifigan(T=8):
    hg = load_by_path("ref_hifigan_modules", os.path.join(REF, "vocoder", "hifigan", "modules", "hifigan.py"))
    for tag, cfg in (("v1", synth.HifiGanConfig()),
                     ("rb2", synth.HifiGanConfig(resblock="2", upsample_rates=(8, 8, 5), upsample_kernel_sizes=(16, 16, 11),
                                                 upsample_initial_channel=128, resblock_kernel_sizes=(3, 5),
                                                 resblock_dilation_sizes=((1, 3), (1, 3))))):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gen = hg.HifiGanGenerator(cfg.as_hparams()).eval()
        sd = synth.make_state_dict(synth.hifigan_shapes(cfg), SEED + 2)
        ref = gen.state_dict()
        assert set(ref) == set(sd), set(ref) ^ set(sd)
        for k in sd:
            assert tuple(ref[k].shape) == tuple(sd[k].shape), (k, ref[k].shape, sd[k].shape)
        gen.load_state_dict(sd)
        mel = torch.from_numpy(prng.uniform(prng.key_seed(SEED, "hg_mel"), 80 * T, -5.0, 1.5).reshape(1, 80, T))
        with torch.no_grad():
            wav = gen(mel)
        np.savez_compressed(os.path.join(GOLD, f"hifigan_{tag}.npz"), mel=mel.numpy(), wav=wav.numpy())
        print("hifigan", tag, wav.shape, float(wav.abs().max()))


def digest(a, n_samples=32):
    """Size-independent fingerprint of a tensor: shape, sum, L2, max-abs and n_samples elements at fixed strided positions."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = (np.arange(n_samples, dtype=np.int64) * 2654435761 + 12345) % a.size
    return {"shape_numel": np.array([a.size], dtype=np.int64), "sum": np.array([a.sum()]), "l2": np.array([np.sqrt((a * a).sum())]),
            "maxabs": np.array([np.abs(a).max()]), "idx": idx, "val": a[idx]}


def gen_fullsize():
    """BASELINE geometry (one 20 s clip: T = 752, L = 80, T_mel = 1504) through the REFERENCE's own modules; only digests are
    committed (sum / L2 / max / 32 sampled elements per tensor) - they catch size-dependent bugs the tiny goldens cannot."""
    from ldm.models.autoencoder1d import AutoencoderKL
    from ldm.modules.diffusionmodules.vocal2music_moe import TxtFlagLargeImprovedDiTV2
    out = {}
    B, T, L, E = 1, 752, 80, 4
    cfg = synth.DiTConfig(num_experts=E)
    net = TxtFlagLargeImprovedDiTV2(in_channels=cfg.in_channels, context_dim=cfg.context_dim, hidden_size=cfg.hidden_size,
                                    depth=cfg.depth, num_heads=cfg.num_heads, max_len=cfg.max_len, num_experts=E,
                                    ori_dim=cfg.ori_dim).eval()
    net.load_state_dict(synth.make_state_dict(synth.dit_shapes(cfg), SEED), strict=True)
    inp = clip_batch(B, T, L)
    t_idx = torch.tensor([583] * B, dtype=torch.long)
    nq = NoiseQueue()
    for blk in block_noise(B, T, E, nfe=0, depth=cfg.depth):
        for a in blk:
            nq.push(a)
    ctx = {"c_concat": {"midi": inp["midi"], "beats": inp["beats"]}, "c_crossattn": inp["t5_cond"], "name": ["x"] * B}
    with torch.no_grad():
        v, _ = net(inp["x_latent"], t_idx, ctx)
    assert len(nq.q) == 0
    nq.restore()
    for k, val in digest(v.numpy()).items():
        out["dit_v_" + k] = val
    out["dit_t_idx"] = t_idx.numpy()
    # VAE decode of a full-length latent, HiFi-GAN of the resulting mel
    vcfg = synth.VAEConfig()
    dd = dict(double_z=True, in_channels=80, out_ch=80, z_channels=20, kernel_size=5, ch=384, ch_mult=[1, 2, 4],
              num_res_blocks=2, attn_layers=[3], down_layers=[0], dropout=0.0)
    ae = AutoencoderKL(embed_dim=20, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}).eval()
    ae.load_state_dict(synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1), strict=False)
    ae.load_state_dict(synth.make_state_dict(synth.vae_encoder_shapes(vcfg), SEED + 3), strict=False)
    z = torch.from_numpy(prng.normal(prng.key_seed(SEED, "full_z"), 20 * T).reshape(1, 20, T))
    with torch.no_grad():
        mel = ae.decode(z)
        mom = ae.encode(mel).parameters
    for k, val in digest(mel.numpy()).items():
        out["vae_mel_" + k] = val
    for k, val in digest(mom.numpy()).items():
        out["vae_moments_" + k] = val
    hg = load_by_path("ref_hifigan_modules", os.path.join(REF, "vocoder", "hifigan", "modules", "hifigan.py"))
    hcfg = synth.HifiGanConfig()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gen = hg.HifiGanGenerator(hcfg.as_hparams()).eval()
    gen.load_state_dict(synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2))
    with torch.no_grad():
        wav = gen(mel)
    for k, val in digest(wav.numpy()).items():
        out["voc_wav_" + k] = val
    np.savez_compressed(os.path.join(GOLD, "fullsize_digests.npz"), **out)
    print("fullsize", v.shape, mel.shape, mom.shape, wav.shape)


def gen_t5():
    """T5 text encoder.  The reference calls transformers.T5EncoderModel (modules.py:197-221); transformers is not
    pinned by requirements.txt - the fixtures pin the behaviour of the version installed here (recorded in the file)."""
    import transformers
    from transformers import T5Config, T5EncoderModel

    def run(cfg, B, L, seed_tag):
        hf = T5EncoderModel(T5Config(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff, num_layers=cfg.num_layers,
                                     num_heads=cfg.num_heads, relative_attention_num_buckets=cfg.relative_attention_num_buckets,
                                     relative_attention_max_distance=cfg.relative_attention_max_distance, dropout_rate=0.0,
                                     layer_norm_epsilon=cfg.layer_norm_epsilon, feed_forward_proj="gated-gelu", tie_word_embeddings=False)).eval()
        sd = synth.make_state_dict(synth.t5_encoder_shapes(cfg), SEED + 5)
        full = dict(sd)
        full["encoder.embed_tokens.weight"] = sd["shared.weight"]
        assert set(hf.state_dict()) == set(full), set(hf.state_dict()) ^ set(full)
        hf.load_state_dict(full)
        ids = torch.from_numpy((prng.uniform(prng.key_seed(SEED, seed_tag), B * L, 0.0, 1.0) * cfg.vocab_size).astype(np.int64).reshape(B, L))
        ids = ids.clamp(0, cfg.vocab_size - 1)
        ids[:, L - L // 4:] = 0                           # trailing pad tokens, as the tokenizer's padding="max_length" produces
        with torch.no_grad():
            out = hf(input_ids=ids).last_hidden_state
        return ids, out

    small = synth.T5Config(vocab_size=512, num_layers=2)
    ids, out = run(small, 2, 80, "t5_ids_small")
    o = {"ids": ids.numpy(), "out_slice": out[:, :, :48].numpy(), "version": np.array([int(x) for x in transformers.__version__.split(".")[:2]])}
    for k, v in digest(out.numpy()).items():
        o["out_" + k] = v
    full = synth.T5Config(vocab_size=2048)                # all 24 layers at full width; only the vocabulary table is cut down
    ids2, out2 = run(full, 1, 80, "t5_ids_full")
    o["ids24"] = ids2.numpy()
    for k, v in digest(out2.numpy()).items():
        o["out24_" + k] = v
    np.savez_compressed(os.path.join(GOLD, "t5_encode.npz"), **o)
    print("t5", out.shape, out2.shape, float(out.abs().max()), float(out2.abs().max()))


def gen_bigvgan(T=8):
    """BigVGAN generator through the reference's own vocoder/bigvgan/models.py BigVGAN."""
    import vocoder.bigvgan.models as M
    import warnings
    out = {}
    for tag, cfg in (("amp1", synth.BigVGANConfig(upsample_initial_channel=128)),
                     ("amp2", synth.BigVGANConfig(resblock="2", upsample_rates=(8, 8, 5), upsample_kernel_sizes=(16, 16, 11),
                                                  upsample_initial_channel=64, resblock_kernel_sizes=(3, 5),
                                                  resblock_dilation_sizes=((1, 3), (1, 3)), activation="snake", snake_logscale=False))):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = M.BigVGAN(types.SimpleNamespace(**cfg.as_hparams())).eval()
        sd = synth.make_state_dict(synth.bigvgan_shapes(cfg), SEED + 7)
        ref = {k for k in net.state_dict() if not k.endswith("filter")}
        assert ref == set(sd), ref ^ set(sd)
        for k in sd:
            assert tuple(net.state_dict()[k].shape) == tuple(sd[k].shape), k
        net.load_state_dict(sd, strict=False)
        mel = torch.from_numpy(prng.uniform(prng.key_seed(SEED, "bv_mel"), 80 * T, -5.0, 1.5).reshape(1, 80, T))
        with torch.no_grad():
            wav = net(mel)
        out[f"{tag}_mel"] = mel.numpy()
        out[f"{tag}_wav"] = wav.numpy()
        print("bigvgan", tag, wav.shape, float(wav.abs().max()), float(wav.std()))
    np.savez_compressed(os.path.join(GOLD, "bigvgan.npz"), **out)


def gen_melnet():
    """Log-mel front-end through the reference's own preprocess/NAT_mel.py MelNet.  librosa is absent here: its
    `filters.mel` import is served by the oracle's restatement (oracle/ref_cpu.py slaney_mel_filterbank), so the fixture pins the
    clamp / reflect-pad / STFT / magnitude / log10 arithmetic of MelNet.forward, not the filterbank itself."""
    fm = _mod("librosa.filters", mel=lambda sr, n_fft, n_mels, fmin, fmax: R.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
    _mod("librosa", filters=fm)
    nat = load_by_path("ref_nat_mel", os.path.join(REF, "preprocess", "NAT_mel.py"))
    hp = dict(fft_size=1280, audio_num_mel_bins=80, audio_sample_rate=24000, hop_size=320, win_size=1280, fmin=0, fmax=8000)   # mel_spec_24k.py:300-307
    net = nat.MelNet(hp)
    out = {"hp_keys": np.array(sorted(hp)), "hp_vals": np.array([hp[k] for k in sorted(hp)], dtype=np.int64)}
    # (a) noise louder than full scale (exercises the clamp), two clips; (b) a quiet two-tone clip with a silent tail (log floor);
    # (c) a length that is not a multiple of the hop
    a = prng.normal(SEED + 900, 2 * 20 * 320).reshape(2, -1) * 0.6
    t = np.arange(24 * 320, dtype=np.float64) / 24000.0
    b = (0.2 * np.sin(2 * np.pi * 440.0 * t) + 0.05 * np.sin(2 * np.pi * 3000.0 * t)).astype(np.float32)[None]
    b[:, 16 * 320:] = 0.0
    c = prng.normal(SEED + 901, 13 * 320 + 123).reshape(1, -1) * 0.1
    for tag, wav in (("a", a), ("b", b), ("c", c)):
        out[f"wav_{tag}"] = wav.astype(np.float32)
        out[f"mel_{tag}"] = net(torch.from_numpy(wav.astype(np.float32))).numpy()
    out["mel_basis"] = net.mel_basis.numpy()
    np.savez_compressed(os.path.join(GOLD, "melnet.npz"), **out)
    print("melnet", {k: v.shape for k, v in out.items()})


def use_reference_paths():
    """Make `import ldm...` / `vocoder...` / `utils...` resolve to the REFERENCE: its packages have no __init__.py (namespace
    packages), so the build's same-named shim packages would win wherever they sit on sys.path.  versband_amd is already
    imported; the repo root (and the cwd entry) leave sys.path and any cached shim modules are dropped."""
    here = {os.path.abspath(p) for p in (REPO, os.getcwd())}
    sys.path[:] = [REF] + [p for p in sys.path if p and os.path.abspath(p) not in here and p != REF]
    for m in [m for m in sys.modules if m.split(".")[0] in ("ldm", "vocoder", "utils", "preprocess")]:
        if not getattr(sys.modules[m], "__vb_stub__", False):
            del sys.modules[m]


def main():
    assert os.path.isdir(REF), "the reference is only available in the build container"
    os.makedirs(GOLD, exist_ok=True)
    install_stubs()
    use_reference_paths()
    torch.set_grad_enabled(False)
    gen_dit(4, "e4")
    gen_dit(8, "e8", B=1, T=16, L=8)
    gen_vae()
    gen_vae_encode()
    gen_hifigan()
    gen_sampler()
    gen_fullsize()
    gen_bigvgan()
    gen_t5()
    gen_melnet()


if __name__ == "__main__":
    main()
