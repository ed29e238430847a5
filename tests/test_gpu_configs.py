"""GPU coverage of the remaining BASELINE.json configs and of the CLI:
 configs[2]  Band-MoE stress (num_experts=8, full-size T=752) vs the oracle, routing bit-exact;
 configs[4]  long-form: chunked latent sampling + overlap-add vocoder (build-defined);
 configs[0]/harness  scripts/test_final.py end to end on synthetic items."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from tests.helpers import SEED, clip_batch, describe, exp_noise, gumbel_arrays, rel_l2
from versband_amd import longform
from versband_amd import model as vm
from versband_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from versband_amd.engine import Context
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return Context("cuda:0")


def test_c1_one_clip_ten_steps_vs_oracle(ctx):
    """BASELINE configs[0] - the reference's own CPU-runnable case: ONE 20 s clip (T = 752, L = 80), 10 Euler steps with CFG,
    VAE decode; the HIP sampler (split precision, injected routing noise) against the pinned CPU oracle at full geometry."""
    from tests.helpers import gumbel_arrays_steps
    from versband_amd.engine import DiTEngine, build_vae_decoder
    dcfg = synth.DiTConfig()
    sd = synth.make_state_dict(synth.dit_shapes(dcfg), SEED)
    sdv = synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), SEED + 1)
    B, T, Lc, E, steps, scale = 1, 752, 80, 4, 10, 3.0
    inp = clip_batch(B, T, Lc)
    noise_steps = [[exp_noise(B, T, E, 2 * k + br, 4) for br in (0, 1)] for k in range(steps)]
    eng = DiTEngine(ctx, dcfg, sd, precision="split")
    cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"], inp["beats"], T)
    idx, dts = vm.euler_tables(steps + 1)
    assert list(idx) == [0, 100, 200, 300, 400, 500, 600, 700, 800, 900]           # SURVEY 8a A3 golden indices
    z = eng.sample_cfg(inp["x_latent"], cond, idx, dts, scale, noise=gumbel_arrays_steps(noise_steps))
    mel = build_vae_decoder(ctx, sdv).run(z)
    torch.cuda.synchronize()
    cc = ref_cpu.dit_precompute(sd, inp["t5_cond"], inp["midi"], inp["beats"], T)
    cu = ref_cpu.dit_precompute(sd, inp["t5_uncond"], inp["midi"], inp["beats"], T)
    z_ref = ref_cpu.sample_cfg(sd, inp["x_latent"], cc, cu, scale, steps + 1, lambda k, br: noise_steps[k][br])
    mel_ref = ref_cpu.vae_decode(sdv, z_ref)
    assert rel_l2(z, z_ref) < 1e-3, describe("10-step latent vs oracle", z, z_ref)
    assert float((mel.cpu() - mel_ref).abs().mean()) < 1e-3
    # the product's default VAE / vocoder arithmetic (round 6: fp32 with F(2,3) minimal filtering) on the same latent, through to the waveform:
    # north_star's bounds (mel L1 < 1e-3, 1e-3 relative) against the oracle - measured at fp32 roundoff, like the direct fp32 kernels
    from versband_amd.engine import build_hifigan
    hcfg = synth.HifiGanConfig()
    sdh = synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2)
    wav_ref = ref_cpu.hifigan_forward(sdh, hcfg.as_hparams(), mel_ref)
    for prec in ("fp32mf", "fp32"):
        melp = build_vae_decoder(ctx, sdv, precision=prec).run(z)
        wavp = build_hifigan(ctx, sdh, hcfg.as_hparams(), precision=prec).run(melp)
        torch.cuda.synchronize()
        l1, rw = float((melp.cpu() - mel_ref).abs().mean()), rel_l2(wavp, wav_ref)
        print(f"C1 {prec}: mel L1 {l1:.3e}, wav rel-L2 {rw:.3e}")
        assert l1 < 1e-3 and rw < 1e-3, f"{prec}: mel L1 {l1:.3e}, wav rel-L2 {rw:.3e}"


def _assert_routes_equal_up_to_ties(got, ref_idx, logits, exp_draws, what, tie=2e-5):
    """Hard Gumbel routing must reproduce the oracle's index bit for bit on identical noise.  The only admissible difference: a token
    whose two best values of (logit + Gumbel) in the ORACLE are closer than the fp32 evaluation error of a logit (a 768-term dot
    product evaluated in a different - mathematically equal - order: folded caption gate, bf16x3 products) - there the argmax is
    not defined to fp32 accuracy by the reference either.  Every differing token is checked to be such a near-tie, and the
    chosen index must be the oracle's runner-up."""
    bad = (got != ref_idx).nonzero().squeeze(1)
    if bad.numel() == 0:
        return
    y = (logits.double() - exp_draws.double().log())[bad]
    top2 = y.topk(2, dim=1)
    margin = (top2.values[:, 0] - top2.values[:, 1]) / top2.values[:, 0].abs().clamp_min(1.0)
    assert bool((margin < tie).all()), f"{what}: {bad.numel()} routes differ and not all are near-ties (worst margin {float(margin.max()):.2e})"
    assert torch.equal(got[bad], top2.indices[:, 1]), f"{what}: a differing route is not the oracle's runner-up"
    assert bad.numel() <= 1e-3 * got.numel(), f"{what}: {bad.numel()} near-ties of {got.numel()} decisions"


def test_c3_moe_stress_e8_full_size_vs_oracle(ctx):
    """num_experts = 8 (band = 96 channels: K tail of the band GEMMs, 16 routed groups), T = 752, L = 80."""
    from versband_amd.engine import DiTEngine
    cfg = synth.DiTConfig(num_experts=8)
    sd = synth.make_state_dict(synth.dit_shapes(cfg), SEED)
    eng = DiTEngine(ctx, cfg, sd, precision="split")
    B, T, Lc, E = 2, 752, 80, 8
    inp = clip_batch(B, T, Lc)
    noise = [exp_noise(B, T, E, br, 4) for br in (0, 1)]
    cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"], inp["beats"], T)
    t_idx = torch.full((2 * B,), 583, dtype=torch.long)
    v, routes = eng.forward(inp["x_latent"], t_idx, cond, noise=gumbel_arrays(noise), return_routes=True)
    torch.cuda.synchronize()
    N = B * T
    for br, key in ((0, "t5_cond"), (1, "t5_uncond")):
        c = ref_cpu.dit_precompute(sd, inp[key], inp["midi"], inp["beats"], T)
        ref, aux = ref_cpu.dit_forward(sd, inp["x_latent"], t_idx[:B], c, noise[br], return_aux=True)
        assert rel_l2(v[br * B:(br + 1) * B], ref) < 1e-4, describe(f"E=8 forward branch {br}", v[br * B:(br + 1) * B], ref)
        for i in range(4):
            got_c = routes[i, 0, br * N:(br + 1) * N].cpu().long()
            got_a = routes[i, 1, br * N:(br + 1) * N].cpu().long()
            _assert_routes_equal_up_to_ties(got_c, aux[f"ic{i}"], aux[f"lc{i}"], noise[br][i][1], f"block {i} caption gate")
            _assert_routes_equal_up_to_ties(got_a, aux[f"ia{i}"], c[f"la{i}"].reshape(N, E), noise[br][i][2], f"block {i} acoustic gate")
        hist = torch.bincount(routes[0, 0].cpu().long(), minlength=E)
        assert (hist > 0).all(), f"degenerate routing {hist.tolist()}"


def test_c5_longform_chunked_sampling(ctx):
    from versband_amd.engine import DiTEngine
    cfg = synth.DiTConfig()
    sd = synth.make_state_dict(synth.dit_shapes(cfg), SEED)
    eng = DiTEngine(ctx, cfg, sd, precision="bf16")
    B, T, Lc, win, ov = 2, 80, 8, 48, 16
    inp = clip_batch(B, T, Lc)
    idx, dts = vm.euler_tables(4)
    z = longform.sample_long(eng, inp["x_latent"], inp["t5_cond"], inp["t5_uncond"], inp["midi"], inp["beats"], idx, dts, 3.0,
                             window=win, overlap=ov, seed=3, clip_base=0)
    torch.cuda.synchronize()
    assert z.shape == (B, 20, T) and torch.isfinite(z).all()
    plan = longform.plan_windows(T, win, ov)
    # window 0 sampled on its own (same clip keys) reproduces the chunked result outside the overlap
    s0, n0 = plan[0]
    cond0 = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"][..., 2 * s0:2 * (s0 + n0)],
                                inp["beats"][..., 2 * s0:2 * (s0 + n0)], n0)
    z0 = eng.sample_cfg(inp["x_latent"][:, :, s0:s0 + n0], cond0, idx, dts, 3.0, seed=3, clip_base=0)
    torch.cuda.synchronize()
    keep = plan[1][0]
    assert torch.equal(z[:, :, :keep].cpu(), z0[:, :, :keep].cpu()), describe("window interior", z[:, :, :keep], z0[:, :, :keep])
    # a clip that fits one window goes through untouched
    zs = longform.sample_long(eng, inp["x_latent"][:, :, :40], inp["t5_cond"], inp["t5_uncond"], inp["midi"][..., :80],
                              inp["beats"][..., :80], idx, dts, 3.0, window=win, overlap=ov, seed=3)
    cond1 = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"][..., :80], inp["beats"][..., :80], 40)
    z1 = eng.sample_cfg(inp["x_latent"][:, :, :40], cond1, idx, dts, 3.0, seed=3)
    torch.cuda.synchronize()
    assert torch.equal(zs.cpu(), z1.cpu())
    # past max_len the plain path refuses loudly instead of reading beyond the RoPE table
    from versband_amd._lib import VersbandError
    big = clip_batch(1, 1504, Lc)
    cb = eng.precompute_cond(torch.cat([big["t5_cond"], big["t5_uncond"]]), big["midi"], big["beats"], 1504)
    with pytest.raises(VersbandError):
        eng.sample_cfg(big["x_latent"], cb, idx, dts, 3.0)


def test_c5_crossfade_kernel_equals_torch_restatement(ctx):
    """vb_crossfade_windows (the product path of sample_long) against longform.crossfade_windows, the torch restatement the oracle's
    long-form fixture was generated with: same windows, same ramps - equal to fp32 rounding of the ramp weights, exact where one window
    covers a sample alone; a window list that does not cover [0, T) is refused."""
    from versband_amd import _lib as L
    B, C, T, win, ov = 3, 20, 4500, 1500, 128
    plan = longform.plan_windows(T, win, ov)
    zw = torch.from_numpy(synth.prng.normal(41, len(plan) * B * C * win).reshape(len(plan) * B, C, win)).cuda()
    got = longform.crossfade_windows_hip(ctx.lib, zw, plan, B, T)
    want = longform.crossfade_windows([zw[i * B:(i + 1) * B] for i in range(len(plan))], plan, T)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (B, C, T)
    assert float((got - want).abs().max()) <= 2e-7 * float(want.abs().max()), describe("crossfade", got, want)
    assert torch.equal(got[:, :, :plan[1][0]], zw[:B, :, :plan[1][0]])           # window 0 alone: untouched
    with pytest.raises(L.VersbandError):
        longform.crossfade_windows_hip(ctx.lib, zw, [plan[0], (win + 5, win)] + list(plan[2:]), B, T)   # samples win .. win + 4 uncovered


def test_c5_overlap_add_vocoder_equals_whole_clip(ctx):
    from versband_amd.engine import build_hifigan
    hcfg = synth.HifiGanConfig()
    sdh = synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2)
    net = build_hifigan(ctx, sdh, hcfg.as_hparams())
    mel = torch.from_numpy(synth.prng.uniform(9, 2 * 80 * 700, -5.0, 1.5).reshape(2, 80, 700)).cuda()
    whole = net.run(mel)
    chunked = longform.vocode_chunked(net, mel, chunk=256, halo=32)
    torch.cuda.synchronize()
    assert chunked.shape == whole.shape
    assert float((chunked - whole).abs().max()) < 2e-6, describe("chunked vs whole vocoding", chunked, whole)


def test_cli_synthetic_end_to_end(tmp_path):
    out = tmp_path / "gen"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "test_final.py"), "--synthetic", "2", "--synthetic_frames", "150",
           "--ddim_steps", "3", "--scales", "3", "--n_samples", "2", "--save_dir", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = out / "cond_gtcodec_accomp_scale_3.0"
    wavs = sorted(os.listdir(d))
    assert wavs == ["0-0000[0][accomp].wav", "0-0000[1][accomp].wav", "0-0001[0][accomp].wav", "0-0001[1][accomp].wav"]
    import wave
    with wave.open(str(d / wavs[0])) as f:
        assert f.getframerate() == 24000 and f.getsampwidth() == 2 and f.getnframes() == 152 * 320
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").astype(np.float64) / 32767.0
    rms_db = 20 * np.log10(np.sqrt(np.mean(pcm ** 2)))
    assert abs(rms_db + 23.0) < 0.5                      # normalize_loudness(-23 dB RMS), scripts/test_final.py:342-347
    assert (out / "clap.csv").exists()


@pytest.mark.gpu
def test_bigvgan_reference_api(tmp_path):
    """VocoderBigVGAN(ckpt_dir)(mel) through the reference-named class and module path (best_netG.pt + args.yml on disk)."""
    import yaml
    from oracle import ref_cpu
    from vocoder.bigvgan.models import VocoderBigVGAN
    cfg = synth.BigVGANConfig(upsample_initial_channel=64)
    sd = synth.make_state_dict(synth.bigvgan_shapes(cfg), 77)
    torch.save({"generator": sd}, tmp_path / "best_netG.pt")
    (tmp_path / "args.yml").write_text(yaml.safe_dump(cfg.as_hparams()))
    voc = VocoderBigVGAN(str(tmp_path), device="cuda:0")
    mel = synth.prng.uniform(3, 80 * 21, -5.0, 1.5).reshape(80, 21).astype(np.float32)
    wav = voc(mel)
    ref = ref_cpu.bigvgan_forward(sd, cfg.as_hparams(), torch.from_numpy(mel)[None]).view(-1).numpy()
    assert wav.shape == ref.shape == (21 * 320,)
    assert np.linalg.norm(wav - ref) / np.linalg.norm(ref) < 3e-4


def test_c3_moe_stress_e8_batch32_at_size(ctx):
    """BASELINE configs[2] AT its workload: num_experts = 8 (band = 96), batch 32 -> 48 128 token rows per CFG branch in one
    evaluation (bf16 production precision, the grouped-GEMM path bench.py --workload c3 times).  Router noise is keyed by the
    global clip index, so three clips of the batch (first, middle, last) are replayed one by one by the CPU oracle on the same
    device-drawn noise: relative error at the bf16 tolerance, routing equal up to near-ties; every expert of every group is used."""
    from versband_amd import prng
    from versband_amd.engine import DiTEngine
    cfg = synth.DiTConfig(num_experts=8)
    sd = synth.make_state_dict(synth.dit_shapes(cfg), SEED)
    eng = DiTEngine(ctx, cfg, sd, precision="bf16")
    B, T, Lc, E, seed, nfe = 32, 752, 80, 8, 77, 5
    inp = clip_batch(B, T, Lc)
    cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"], inp["beats"], T)
    t_idx = torch.full((2 * B,), 583, dtype=torch.long)
    v, routes = eng.forward(inp["x_latent"], t_idx, cond, seed=seed, clip_base=0, nfe=nfe, return_routes=True)
    torch.cuda.synchronize()
    assert torch.isfinite(v).all()
    N = B * T
    for grp in (0, 1):
        for i in range(4):
            hist = torch.bincount(routes[i, grp].cpu().long(), minlength=E)
            assert (hist > 0).all() and int(hist.sum()) == 2 * N, f"block {i} group {grp}: degenerate routing {hist.tolist()}"
    for clip in (0, 17, 31):
        for br, key in ((0, "t5_cond"), (1, "t5_uncond")):
            noise = [tuple(torch.from_numpy(prng.device_router_exponentials(seed, clip, nfe, br, blk, gate, T, w))
                           for gate, w in ((0, 2), (1, E), (2, E))) for blk in range(4)]
            c = ref_cpu.dit_precompute(sd, inp[key][clip:clip + 1], inp["midi"][clip:clip + 1], inp["beats"][clip:clip + 1], T)
            ref, aux = ref_cpu.dit_forward(sd, inp["x_latent"][clip:clip + 1], t_idx[:1], c, noise, return_aux=True)
            got = v[br * B + clip:br * B + clip + 1]
            assert rel_l2(got, ref) < 2e-3, describe(f"E=8 B=32 clip {clip} branch {br}", got, ref)
            r0 = br * N + clip * T
            flips = sum(int((routes[i, g, r0:r0 + T].cpu().long() != aux[("ic", "ia")[g] + str(i)]).sum()) for i in range(4) for g in (0, 1))
            assert flips <= 2, f"clip {clip} branch {br}: {flips} of {8 * T} routes differ from the oracle in bf16 precision"


def test_c5_longform_at_size(ctx):
    """BASELINE configs[4] AT its workload: one 120 s clip = 4500 latent frames sampled as 4 windows of max_len = 1500 tokens (one
    batch through vb_sample_cfg), cross-faded, VAE-decoded as a whole (9000 mel frames) and vocoded in halo'd chunks."""
    from versband_amd.engine import DiTEngine, build_hifigan, build_vae_decoder
    cfg = synth.DiTConfig()
    sd = synth.make_state_dict(synth.dit_shapes(cfg), SEED)
    eng = DiTEngine(ctx, cfg, sd, precision="bf16")
    B, T, Lc, steps = 1, 4500, 80, 6
    inp = clip_batch(B, T, Lc)
    idx, dts = vm.euler_tables(steps + 1)
    plan = longform.plan_windows(T, cfg.max_len, 128)
    assert plan == [(0, 1500), (1372, 1500), (2744, 1500), (3000, 1500)]
    z = longform.sample_long(eng, inp["x_latent"], inp["t5_cond"], inp["t5_uncond"], inp["midi"], inp["beats"], idx, dts, 3.0,
                             window=cfg.max_len, overlap=128, seed=3, clip_base=0)
    torch.cuda.synchronize()
    assert z.shape == (B, 20, T) and torch.isfinite(z).all()
    # window 0 sampled on its own (same clip keys) reproduces the chunked result outside the overlap, bitwise
    n0 = plan[0][1]
    cond0 = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"][..., :2 * n0], inp["beats"][..., :2 * n0], n0)
    z0 = eng.sample_cfg(inp["x_latent"][:, :, :n0], cond0, idx, dts, 3.0, seed=3, clip_base=0)
    torch.cuda.synchronize()
    keep = plan[1][0]
    assert torch.equal(z[:, :, :keep].cpu(), z0[:, :, :keep].cpu()), describe("window interior", z[:, :, :keep], z0[:, :, :keep])
    # inside the first overlap [1372, 1500) the result is the convex combination w * z_next + (1 - w) * z_prev of the two windows' own
    # latents, w the linear ramp of crossfade_windows (window 1 = batch row 1 of the chunked call, so its clip key is clip_base 1)
    s1, n1 = plan[1]
    cond1 = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"][..., 2 * s1:2 * (s1 + n1)],
                                inp["beats"][..., 2 * s1:2 * (s1 + n1)], n1)
    z1 = eng.sample_cfg(inp["x_latent"][:, :, s1:s1 + n1].contiguous(), cond1, idx, dts, 3.0, seed=3, clip_base=1)
    torch.cuda.synchronize()
    ov = n0 - keep
    assert ov == 128
    wgt = torch.linspace(0, 1, ov + 2, dtype=torch.float64)[1:-1]
    zp, zn, zo = z0[:, :, keep:n0].cpu().double(), z1[:, :, :ov].cpu().double(), z[:, :, keep:n0].cpu().double()
    expect = wgt * zn + (1.0 - wgt) * zp
    err = ((zo - expect).abs() / expect.abs().clamp_min(1.0)).max()
    assert float(err) <= 2e-7, f"overlap is not w * z_next + (1 - w) * z_prev: {float(err):.3e}"
    assert bool(((zo >= torch.minimum(zp, zn) - 1e-6) & (zo <= torch.maximum(zp, zn) + 1e-6)).all())
    # and past the overlap, up to the next one, the chunked result IS window 1's own latent
    nxt = plan[2][0]
    assert torch.equal(z[:, :, n0:nxt].cpu(), z1[:, :, n0 - s1:nxt - s1].cpu())
    sdv = synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), SEED + 1)
    mel = build_vae_decoder(ctx, sdv).run(z)
    assert mel.shape == (B, 80, 2 * T) and torch.isfinite(mel).all()
    hcfg = synth.HifiGanConfig()
    net = build_hifigan(ctx, synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2), hcfg.as_hparams())
    chunked = longform.vocode_chunked(net, mel, chunk=3000, halo=32)
    whole = net.run(mel)
    torch.cuda.synchronize()
    assert chunked.shape == whole.shape == (B, 1, 2 * T * 320)
    assert float((chunked - whole).abs().max()) < 2e-6, describe("chunked vs whole vocoding at 120 s", chunked, whole)
    assert float(whole.abs().max()) <= 1.0
