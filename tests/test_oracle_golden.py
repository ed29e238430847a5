"""Pin the CPU oracle (oracle/ref_cpu.py) against fixtures produced by the REAL
reference (oracle/gen_golden.py, run in the build container).  CPU-only."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu
from tests.helpers import check_digest, clip_batch, exp_noise
from versband_amd import synth

SEED = 1234


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def dit_sd():
    return {E: synth.make_state_dict(synth.dit_shapes(synth.DiTConfig(num_experts=E)), SEED) for E in (4, 8)}


@pytest.mark.parametrize("tag,E", [("e4", 4), ("e8", 8)])
@pytest.mark.parametrize("dense", [False, True])
def test_dit_forward_matches_reference(golden_dir, dit_sd, tag, E, dense):
    g = _load(golden_dir, f"dit_forward_{tag}.npz")
    B, T, L, E_, _ = g["meta"]
    assert E_ == E
    sd = dit_sd[E]
    x = torch.from_numpy(g["x"])
    midi, beats = torch.from_numpy(g["midi"]), torch.from_numpy(g["beats"])
    for br, key in ((0, "t5_cond"), (1, "t5_uncond")):
        cond = ref_cpu.dit_precompute(sd, torch.from_numpy(g[key]), midi, beats, int(T))
        noise = [tuple(torch.from_numpy(g[f"noise{br}_{i}_{j}"]) for j in range(3)) for i in range(4)]
        v = ref_cpu.dit_forward(sd, x, torch.from_numpy(g["t_idx"]), cond, noise, dense=dense)
        assert _rel(v, g[f"v{br}"]) < 2e-6, (br, _rel(v, g[f"v{br}"]))


def test_t_index_tables(golden_dir):
    g = _load(golden_dir, "t_index_tables.npz")
    for n in (10, 24, 50):
        _, idx = ref_cpu.t_index_table(n + 1)
        assert idx == list(g[f"tidx{n}"])
    # the quantisation quirk of SURVEY Q3 is really there
    assert list(g["tidx50"])[5] == 99 and list(g["tidx50"])[10] == 199


def test_vae_decode_matches_reference(golden_dir):
    g = _load(golden_dir, "vae_decode.npz")
    sd = synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), SEED + 1)
    mel = ref_cpu.vae_decode(sd, torch.from_numpy(g["z"]))
    assert _rel(mel, g["mel"]) < 2e-6


def test_vae_encode_matches_reference(golden_dir):
    """Encoder1D + quant_conv + DiagonalGaussianDistribution against the reference's AutoencoderKL.encode."""
    g = _load(golden_dir, "vae_encode.npz")
    sd = synth.make_state_dict(synth.vae_encoder_shapes(synth.VAEConfig()), SEED + 3)
    mom = ref_cpu.vae_encode(sd, torch.from_numpy(g["x"]))
    assert mom.shape == g["moments"].shape
    assert _rel(mom, g["moments"]) < 2e-6
    assert _rel(ref_cpu.gaussian_posterior(mom, torch.from_numpy(g["eps"])), g["z"]) < 2e-6
    assert _rel(ref_cpu.gaussian_posterior(mom), g["mode"]) < 2e-6


@pytest.mark.parametrize("tag", ["v1", "rb2"])
def test_hifigan_matches_reference(golden_dir, tag):
    g = _load(golden_dir, f"hifigan_{tag}.npz")
    cfg = synth.HifiGanConfig() if tag == "v1" else synth.HifiGanConfig(
        resblock="2", upsample_rates=(8, 8, 5), upsample_kernel_sizes=(16, 16, 11), upsample_initial_channel=128,
        resblock_kernel_sizes=(3, 5), resblock_dilation_sizes=((1, 3), (1, 3)))
    sd = synth.make_state_dict(synth.hifigan_shapes(cfg), SEED + 2)
    wav = ref_cpu.hifigan_forward(sd, cfg.as_hparams(), torch.from_numpy(g["mel"]))
    assert wav.shape == g["wav"].shape
    assert _rel(wav, g["wav"]) < 2e-6


def test_sample_cfg_and_decode_match_reference(golden_dir, dit_sd):
    g = _load(golden_dir, "sample_cfg_3step.npz")
    B, T, L, E, seed, steps = [int(v) for v in g["meta"]]
    sd = dit_sd[E]
    midi, beats = torch.from_numpy(g["midi"]), torch.from_numpy(g["beats"])
    cc = ref_cpu.dit_precompute(sd, torch.from_numpy(g["t5_cond"]), midi, beats, T)
    cu = ref_cpu.dit_precompute(sd, torch.from_numpy(g["t5_uncond"]), midi, beats, T)

    def noise_fn(k, br):
        out = []
        for i in range(4):
            out.append(tuple(torch.from_numpy(np.concatenate(
                [synth.gumbel_exponentials(seed, b, 2 * k + br, i, gate, T, w) for b in range(B)], 0))
                for gate, w in ((0, 2), (1, E), (2, E))))
        return out
    z, traj = ref_cpu.sample_cfg(sd, torch.from_numpy(g["x"]), cc, cu, float(g["scale"]), steps + 1, noise_fn,
                                 return_traj=True)
    assert _rel(traj, g["traj"]) < 5e-6
    assert _rel(z, g["z"]) < 5e-6
    sdv = synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), SEED + 1)
    mel = ref_cpu.vae_decode(sdv, z, scale_factor=float(g["scale_factor"]))
    assert _rel(mel, g["mel"]) < 5e-6


def test_router_top1_tie_break_and_weight():
    logits = torch.tensor([[1.0, 1.0, 0.5, 1.0], [0.0, 2.0, 2.0, -1.0]])
    idx, w = ref_cpu.router_top1(logits, torch.zeros_like(logits))
    assert idx.tolist() == [0, 1]
    assert torch.allclose(w, torch.ones(2), atol=2e-7)


def test_fullsize_digests_match_reference(golden_dir, dit_sd):
    """BASELINE geometry (T = 752, L = 80, T_mel = 1504): the oracle against digests of the reference's own outputs."""
    g = _load(golden_dir, "fullsize_digests.npz")
    B, T, Lc, E = 1, 752, 80, 4
    inp = clip_batch(B, T, Lc)
    sd = dit_sd[E]
    cond = ref_cpu.dit_precompute(sd, inp["t5_cond"], inp["midi"], inp["beats"], T)
    v = ref_cpu.dit_forward(sd, inp["x_latent"], torch.from_numpy(g["dit_t_idx"]), cond, exp_noise(B, T, E, 0, 4))
    check_digest(v, g, "dit_v_", 5e-6)
    vcfg = synth.VAEConfig()
    z = torch.from_numpy(synth.prng.normal(synth.prng.key_seed(SEED, "full_z"), 20 * T).reshape(1, 20, T))
    mel = ref_cpu.vae_decode(synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1), z)
    check_digest(mel, g, "vae_mel_", 5e-6)
    mom = ref_cpu.vae_encode(synth.make_state_dict(synth.vae_encoder_shapes(vcfg), SEED + 3), mel)
    check_digest(mom, g, "vae_moments_", 5e-6)
    hcfg = synth.HifiGanConfig()
    wav = ref_cpu.hifigan_forward(synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2), hcfg.as_hparams(), mel)
    check_digest(wav, g, "voc_wav_", 2e-5)


def test_t5_encoder_matches_transformers(golden_dir):
    """the T5 restatement against fixtures produced by transformers.T5EncoderModel (2 layers: output slice + digest;
    all 24 layers: digest)."""
    g = _load(golden_dir, "t5_encode.npz")
    sd = synth.make_state_dict(synth.t5_encoder_shapes(synth.T5Config(vocab_size=512, num_layers=2)), SEED + 5)
    out = ref_cpu.t5_encode(sd, torch.from_numpy(g["ids"]))
    assert _rel(out[:, :, :48], g["out_slice"]) < 2e-6
    check_digest(out, g, "out_", 5e-6)
    sd24 = synth.make_state_dict(synth.t5_encoder_shapes(synth.T5Config(vocab_size=2048)), SEED + 5)
    check_digest(ref_cpu.t5_encode(sd24, torch.from_numpy(g["ids24"])), g, "out24_", 2e-5)


_BV_CFGS = {"amp1": dict(upsample_initial_channel=128),
            "amp2": dict(resblock="2", upsample_rates=(8, 8, 5), upsample_kernel_sizes=(16, 16, 11), upsample_initial_channel=64,
                         resblock_kernel_sizes=(3, 5), resblock_dilation_sizes=((1, 3), (1, 3)), activation="snake", snake_logscale=False)}


@pytest.mark.parametrize("tag", ["amp1", "amp2"])
def test_bigvgan_matches_reference(golden_dir, tag):
    """BigVGAN (AMPBlock1 + SnakeBeta log-scale / AMPBlock2 + Snake) against the reference's own generator."""
    g = _load(golden_dir, "bigvgan.npz")
    cfg = synth.BigVGANConfig(**_BV_CFGS[tag])
    sd = synth.make_state_dict(synth.bigvgan_shapes(cfg), SEED + 7)
    wav = ref_cpu.bigvgan_forward(sd, cfg.as_hparams(), torch.from_numpy(g[tag + "_mel"]))
    assert _rel(wav, g[tag + "_wav"]) < 1e-5


# ---------------------------------------------------------------- log-mel front-end
MEL_HP = dict(fft_size=1280, audio_num_mel_bins=80, audio_sample_rate=24000, hop_size=320, win_size=1280, fmin=0, fmax=8000)


def mel_close(got, ref, log_tol):
    """log10-mel comparison: tight in the linear domain (the quantity the STFT produces), `log_tol` in the log domain - spectral
    leakage bins 5 decades under the peak amplify the 1e-7-relative noise of ANY fp32 transform (the reference's FFT included)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    lin = np.abs(10.0 ** got - 10.0 ** ref).max()
    peak = (10.0 ** ref).max()
    assert lin <= 5e-6 + 2e-6 * peak, f"linear mel error {lin:.3e} (peak {peak:.3g})"
    assert np.abs(got - ref).max() <= log_tol, f"log10-mel error {np.abs(got - ref).max():.3e} > {log_tol}"


def test_melnet_oracle_matches_reference_golden(golden_dir):
    g = _load(golden_dir, "melnet.npz")
    assert dict(zip(g["hp_keys"].tolist(), g["hp_vals"].tolist())) == MEL_HP
    fb = ref_cpu.slaney_mel_filterbank(24000, 1280, 80, 0, 8000)
    assert np.array_equal(fb, g["mel_basis"])
    for tag, tol in (("a", 1e-5), ("b", 5e-3), ("c", 1e-5)):
        out = ref_cpu.melnet_forward(torch.from_numpy(g["wav_" + tag]), MEL_HP, torch.from_numpy(fb))
        mel_close(out.numpy(), g["mel_" + tag], tol)
    assert g["mel_b"][:, :, 20:].max() == -5.0          # silent tail sits on the log floor
    assert g["mel_c"].shape[-1] == g["wav_c"].shape[-1] // 320


def test_mel_filterbank_restatements_agree_and_known_answers():
    """The two independent restatements of librosa.filters.mel (oracle: scalar loops; product: numpy) agree; known answers: the
    librosa documentation's example `mel(sr=22050, n_fft=2048)` prints row 0 as [0., 0.016, 0.032, ...] (3 decimals); Slaney
    normalisation makes every filter integrate to ~1 over frequency; a filter is zero outside its two neighbours' centres."""
    from versband_amd import melnet as M
    for sr, n_fft, n_mels, fmin, fmax in ((24000, 1280, 80, 0, 8000), (22050, 2048, 128, 0, None), (16000, 1024, 40, 55, 7600)):
        a = ref_cpu.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        b = M.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        assert a.shape == b.shape == (n_mels, n_fft // 2 + 1) and a.dtype == b.dtype == np.float32
        assert np.abs(a - b).max() <= 1e-8
        df = sr / n_fft
        area = a.astype(np.float64).sum(1) * df
        wide = area[n_mels // 2:]                       # filters several bins wide: the Riemann sum is close to the integral
        assert np.all(np.abs(wide - 1.0) < 0.08), wide
        assert (a >= 0).all() and (a.sum(1) > 0).all()
    doc = ref_cpu.slaney_mel_filterbank(22050, 2048, 128, 0.0, None)
    assert [round(float(v), 3) for v in doc[0, :3]] == [0.0, 0.016, 0.032]
    assert doc[0, -1] == 0.0 and doc[-1, 0] == 0.0


def test_melnet_dft_weights_as_hop_block_convolution(golden_dir):
    """The product's weight layout (vb_melnet_load): the windowed DFT as a k = n_fft/hop convolution over hop-blocks reproduces the
    reference MelNet on CPU (torch conv1d standing in for the HIP convolution kernel)."""
    from versband_amd import melnet as M
    g = _load(golden_dir, "melnet.npz")
    W = torch.from_numpy(M.dft_weights(1280, 320, 1280))
    assert W.shape == (4, 320, 2 * M.im_offset(1280)) and M.im_offset(1280) == 644
    assert float(W[:, :, 641:644].abs().max()) == 0.0 and float(W[:, :, 644 + 641:].abs().max()) == 0.0
    fb = torch.from_numpy(M.mel_filterbank(24000, 1280, 80, 0, 8000))
    for tag, tol in (("a", 1e-5), ("b", 1e-2), ("c", 1e-5)):
        y = torch.from_numpy(g["wav_" + tag]).clamp(-1, 1)
        y = F.pad(y.unsqueeze(1), [480, 480], mode="reflect").squeeze(1)
        T = y.shape[1] // 320 - 3
        X = y[:, :(T + 3) * 320].reshape(y.shape[0], T + 3, 320).transpose(1, 2)
        sp = F.conv1d(X, W.permute(2, 1, 0).contiguous())
        re, im = sp[:, :641], sp[:, 644:644 + 641]
        mel = torch.log10(torch.clamp(torch.matmul(fb, torch.sqrt(re ** 2 + im ** 2 + 1e-9)), min=1e-5))
        mel_close(mel.numpy(), g["mel_" + tag], tol)
    # short window: centred zero-padding like torch.stft
    w = M.hann_window(800, 1280)
    assert w[:240].max() == 0.0 and w[1040:].max() == 0.0 and abs(w[240 + 400] - 1.0) < 1e-6


def test_bench_checks_fixture_is_consistent_with_the_clip0_fixture(golden_dir):
    """tests/golden/bench_c2_checks.npz (oracle/gen_bench_digest.py --checks): its (clip 0, pass 0) entry is the same oracle replay as
    bench_clip0.npz, bit for bit; every (clip, pass) entry and every teacher-forced step is present and finite"""
    g = _load(golden_dir, "bench_c2_checks.npz")
    g0 = _load(golden_dir, "bench_clip0.npz")
    assert np.array_equal(g["z_c0_p0"], g0["z"]) and np.array_equal(g["mel_c0_p0_val"], g0["mel_val"])
    for c in (0, 4):
        for p in (0, 1):
            z = g[f"z_c{c}_p{p}"]
            assert z.shape == (1, 20, 752) and np.isfinite(z).all() and g[f"mel_c{c}_p{p}_idx"].shape == (256,)
    assert not np.array_equal(g["z_c0_p0"], g["z_c0_p1"]) and not np.array_equal(g["z_c0_p0"], g["z_c4_p0"])
    for k in g["tf_steps"]:
        r = g[f"tf_routes_{int(k)}"]
        assert r.shape == (4, 2, 1504) and r.min() >= 0 and r.max() <= 3
        assert g[f"tf_x_{int(k)}"].shape == (1, 20, 752)
