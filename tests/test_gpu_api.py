"""Reference-API members that the end-to-end test does not reach, each against the CPU oracle:
  LatentDiffusion_audio.apply_model      ddpm_audio.py:443-469 (+ DiffusionWrapper 'hybrid', ddpm.py:1418-1436)
  CFMSampler.stochastic_encode / t_start  cfm1_audio_sampler.py:41-46, 110-111
  encode_first_stage / get_first_stage_encoding   ddpm_audio.py:163-170, 411-412
  HifiGAN.vocode                          vocoder/hifigan/hifigan.py:32-43
through the reference-named classes (configs/vocal2music.yaml -> instantiate_from_config)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import ref_cpu
from tests.helpers import SEED, clip_batch, describe, exp_noise, gumbel_arrays_steps, rel_l2
from versband_amd import _lib as L
from versband_amd import model as vm
from versband_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cfm():
    from ldm.util import instantiate_from_config
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    config = vm.load_config(os.path.join(ROOT, "configs", "vocal2music.yaml"))
    config.model.params["precision"] = "split"
    model = instantiate_from_config(config.model)
    vcfg = synth.VAEConfig()
    sd_d = synth.make_state_dict(synth.dit_shapes(synth.DiTConfig()), SEED)
    sd_v = synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1)
    sd_e = synth.make_state_dict(synth.vae_encoder_shapes(vcfg), SEED + 3)
    sd = {**{"model.diffusion_model." + k: v for k, v in sd_d.items()}, **{"first_stage_model." + k: v for k, v in sd_v.items()},
          **{"first_stage_model." + k: v for k, v in sd_e.items()}, "scale_factor": torch.tensor(0.8)}
    model.load_state_dict(sd, strict=False)
    return dict(model=model.to("cuda:0"), sd_d=sd_d, sd_v=sd_v, sd_e=sd_e)


def _device_gumbel_as_exponentials(B, T, E, seed, nfe, depth=4):
    """the counter-based router draws the library makes for evaluation `nfe` (vb_fill_gumbel), turned back into the Exp(1)
    draws the oracle consumes: per block (E1 [N,2], E2 [N,E], E3 [N,E])"""
    lib = L.load()
    out = []
    for blk in range(depth):
        parts = []
        for gate, w in ((0, 2), (1, E), (2, E)):
            g = torch.zeros(B * T * w, device="cuda")
            L.check(lib.vb_fill_gumbel(L.ptr(g), B, 1, T, w, seed, 0, nfe, blk, gate, L.stream_ptr()), "fill_gumbel")
            parts.append(torch.exp(-g.double()).float().cpu().view(B * T, w))
        out.append(tuple(parts))
    return out


def test_apply_model_vs_oracle(cfm):
    """one conditional evaluation through the reference's apply_model(x, t, cond) -> (v, lb_loss); the router noise of call n is the
    library's counter-based stream for evaluation n (fresh noise per call, like the reference's gumbel_softmax)."""
    model, sd = cfm["model"], cfm["sd_d"]
    B, T, Lc, E = 2, 40, 8, 4
    inp = clip_batch(B, T, Lc)
    ac = {"acoustic": torch.zeros(B, 20, 2 * T), "midi": inp["midi"], "beats": inp["beats"]}
    cond = model.get_learned_conditioning({"caption": inp["t5_cond"], "acoustic": ac, "name": ["a"] * B})
    seed = int(torch.initial_seed()) & 0xFFFFFFFF
    cc = ref_cpu.dit_precompute(sd, inp["t5_cond"], inp["midi"], inp["beats"], T)
    outs = []
    for call, tval in enumerate((417, 1000)):         # 1000 = num_timesteps: outside the tabulated sinusoid rows (computed in-kernel)
        t = torch.full((B,), tval, dtype=torch.long)
        nfe = model._nfe
        v, lb = model.apply_model(inp["x_latent"].cuda(), t.cuda(), cond)
        torch.cuda.synchronize()
        assert v.shape == (B, 20, T) and lb.shape == () and float(lb) == 0.0
        ref = ref_cpu.dit_forward(sd, inp["x_latent"], t, cc, _device_gumbel_as_exponentials(B, T, E, seed, nfe))
        assert rel_l2(v, ref) < 1e-3, describe(f"apply_model call {call} (t={tval})", v, ref)
        outs.append(v.cpu())
    assert model._nfe == nfe + 1
    with pytest.raises(NotImplementedError):
        model.apply_model(inp["x_latent"].cuda(), t.cuda(), [cond])


def test_t_start_and_stochastic_encode_vs_oracle(cfm):
    """editing path: mel -> encode_first_stage -> get_first_stage_encoding -> stochastic_encode(t) -> sample_cfg(t_start=k)."""
    from ldm.models.diffusion.cfm1_audio_sampler import CFMSampler
    model, sd, sde = cfm["model"], cfm["sd_d"], cfm["sd_e"]
    sampler = CFMSampler(model, num_timesteps=1000)
    B, T, Lc, E, timesteps, t_start, scale = 2, 24, 8, 4, 7, 3, 3.0
    inp = clip_batch(B, T, Lc)
    mel = torch.from_numpy(synth.prng.uniform(41, B * 80 * 2 * T, -4.0, 0.5).reshape(B, 80, 2 * T))
    post = model.encode_first_stage(mel.cuda())
    eps = torch.from_numpy(synth.prng.normal(42, B * 20 * T).reshape(B, 20, T))
    z0 = model.get_first_stage_encoding(vm.DiagonalGaussianDistribution(post.parameters)).shape        # sampling path runs
    z_enc = float(model.scale_factor) * post.sample(eps)
    mom_ref = ref_cpu.vae_encode(sde, mel)
    z_ref0 = ref_cpu.gaussian_posterior(mom_ref, eps, scale_factor=0.8)
    assert tuple(z0) == (B, 20, T)
    assert rel_l2(z_enc, z_ref0) < 1e-3, describe("encode_first_stage -> encoding", z_enc, z_ref0)
    assert rel_l2(model.get_first_stage_encoding(post.mode()), ref_cpu.gaussian_posterior(mom_ref, None, 0.8)) < 1e-3
    # stochastic_encode (cfm1_audio_sampler.py:41-46): x_t = (1 - t/N) x + (1 - (1 - sigma_min)(1 - t/N)) noise
    tt = torch.tensor([300, 650])
    noise = torch.from_numpy(synth.prng.normal(43, B * 20 * T).reshape(B, 20, T))
    xt = sampler.stochastic_encode(z_ref0, tt, noise=noise)
    tu = 1 - tt.view(B, 1, 1).float() / 1000
    assert torch.allclose(xt, tu * z_ref0 + (1 - (1 - 1e-4) * tu) * noise, atol=0, rtol=0)
    # sample_cfg(t_start=k): the solver integrates t_span[k:] only (:110-111)
    ac = {"acoustic": torch.zeros(B, 20, 2 * T), "midi": inp["midi"], "beats": inp["beats"]}
    c = model.get_learned_conditioning({"caption": inp["t5_cond"], "acoustic": ac, "name": ["a"] * B})
    uc = model.get_learned_conditioning({"caption": inp["t5_uncond"], "acoustic": ac, "name": ["a"] * B})
    steps = timesteps - 1 - t_start
    noise_steps = [[exp_noise(B, T, E, 2 * k + br, 4) for br in (0, 1)] for k in range(steps)]
    z, traj = sampler.sample_cfg(cond=c, unconditional_guidance_scale=scale, unconditional_conditioning=uc, batch_size=B, shape=[20, T],
                                 x_latent=xt, timesteps=timesteps, t_start=t_start, gumbel_noise=gumbel_arrays_steps(noise_steps))
    torch.cuda.synchronize()
    assert traj.shape[0] == steps + 1
    cc = ref_cpu.dit_precompute(sd, inp["t5_cond"], inp["midi"], inp["beats"], T)
    cu = ref_cpu.dit_precompute(sd, inp["t5_uncond"], inp["midi"], inp["beats"], T)
    z_ref = ref_cpu.sample_cfg(sd, xt, cc, cu, scale, timesteps, lambda k, br: noise_steps[k][br], t_start=t_start)
    assert rel_l2(z, z_ref) < 1e-3, describe("sample_cfg(t_start=3) vs oracle", z, z_ref)
    # sample() = no guidance (:49-80)
    z1, _ = sampler.sample(cond=c, batch_size=B, shape=[20, T], x_latent=xt, timesteps=3,
                           gumbel_noise=tuple(g[:, :, :B * T] for g in gumbel_arrays_steps([[noise_steps[k][0]] for k in range(2)])))
    z1_ref = ref_cpu.sample_cfg(sd, xt, cc, None, 1.0, 3, lambda k, br: noise_steps[k][0])
    assert rel_l2(z1, z1_ref) < 1e-3, describe("sample() vs oracle", z1, z1_ref)


def test_hifigan_vocode_and_spec2wav_vs_oracle(tmp_path):
    from vocoder.hifigan import HifiGAN
    hcfg = synth.HifiGanConfig(upsample_initial_channel=128)
    sdh = synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2)
    yaml.safe_dump(hcfg.as_hparams(), open(tmp_path / "config.yaml", "w"))
    torch.save({"state_dict": {"model_gen." + k: v for k, v in sdh.items()}}, tmp_path / "model_ckpt_steps_5.ckpt")   # prefixed-key form
    voc = HifiGAN(vocoder_ckpt=str(tmp_path), device="cuda:0")
    mel = torch.from_numpy(synth.prng.uniform(8, 80 * 33, -5.0, 1.0).reshape(80, 33))
    ref = ref_cpu.hifigan_forward(sdh, hcfg.as_hparams(), mel[None]).view(-1).numpy()
    for arg in (mel, mel.t().contiguous(), mel.numpy()):              # vocode() transposes when dim 1 is not 80 (hifigan.py:36-38)
        wav = voc.vocode(arg)
        assert isinstance(wav, np.ndarray) and wav.dtype == np.float32 and wav.shape == ref.shape == (33 * 320,)
        assert np.linalg.norm(wav - ref) / np.linalg.norm(ref) < 3e-4
    wav = voc.spec2wav(mel.t().contiguous())                          # spec2wav takes [T, 80]
    assert np.linalg.norm(wav - ref) / np.linalg.norm(ref) < 3e-4
    with pytest.raises(AssertionError):
        voc.vocode(mel[None])


def test_host_restatement_of_the_device_noise_stream():
    """prng.device_router_exponentials (host) == the library's counter-based router noise (vb_fill_gumbel / the router kernel's
    gumbel_draw): the integer pipeline is exact, -log(E) agrees with the device Gumbel value to float32 rounding.  This is what
    lets the CPU oracle replay a PRODUCTION run (oracle/gen_bench_digest.py, bench.py's parity check)."""
    from versband_amd import prng
    lib = L.load()
    B, nb, T, W, seed, clip_base, nfe, blk, gate = 3, 2, 257, 4, 1234, 5, 7, 2, 1
    g = torch.zeros(nb * B * T * W, device="cuda")
    L.check(lib.vb_fill_gumbel(L.ptr(g), B, nb, T, W, seed, clip_base, nfe, blk, gate, L.stream_ptr()), "fill_gumbel")
    torch.cuda.synchronize()
    g = g.cpu().view(nb, B, T, W).double()
    for br in range(nb):
        for b in range(B):
            e = prng.device_router_exponentials(seed, clip_base + b, nfe, br, blk, gate, T, W)
            ref = -np.log(e.astype(np.float64))
            err = np.abs(g[br, b].numpy() - ref) / np.maximum(1.0, np.abs(ref))
            assert err.max() < 2e-6, (br, b, float(err.max()))


def test_index_range_check_is_not_skipped_for_converted_inputs():
    """the range check of midi / beats was cached on the address of converted TEMPORARIES (CPU / int32 inputs), which the
    caching allocator recycles - a bad track arriving after a good one of the same size skipped the check and embed_t_kernel read
    out of bounds.  Converted inputs are now checked on every call; resident int64 tracks are cached with a reference held."""
    from versband_amd.engine import Context, DiTEngine
    cfg = synth.DiTConfig()
    eng = DiTEngine(Context("cuda:0"), cfg, synth.make_state_dict(synth.dit_shapes(cfg), SEED), precision="bf16")
    B, T, Lc = 1, 64, 16
    inp = clip_batch(B, T, Lc)
    t5 = torch.cat([inp["t5_cond"], inp["t5_uncond"]])
    for conv in (lambda t: t.to(torch.int32), lambda t: t.clone()):          # CPU int32, CPU int64: both become device temporaries
        eng.precompute_cond(t5, conv(inp["midi"]), conv(inp["beats"]), T)
        torch.cuda.synchronize()
        bad = conv(inp["midi"]).clone()
        bad.view(-1)[3] = 1000
        with pytest.raises(IndexError):
            eng.precompute_cond(t5, bad, conv(inp["beats"]), T)
    # resident tracks: checked once, then served from the cache until they are modified in place
    midi, beats = inp["midi"].to("cuda:0").reshape(B, -1).contiguous(), inp["beats"].to("cuda:0").reshape(B, -1).contiguous()
    eng.precompute_cond(t5, midi, beats, T)
    n = len(eng._checked)
    eng.precompute_cond(t5, midi, beats, T)
    assert len(eng._checked) == n >= 1
    midi[0, 5] = 777
    with pytest.raises(IndexError):
        eng.precompute_cond(t5, midi, beats, T)
