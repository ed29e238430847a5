"""Parity of the BENCHMARKED configuration (BASELINE configs[1]): bf16 production precision, full 20 s geometry
(T = 752, L = 80), many CFG Euler steps, against the pinned CPU oracle at north_star's tolerances - latent relative
error <= 1e-3 and mel L1 < 1e-3 - with the routing flip count of every block reported per step.

Reference semantics under test: flag_large_dit_moe.py:353-386 (fp32 attention path), vocal2music_moe.py:81-93,150-151
(hard Gumbel routing), cfm1_audio.py:154-162 (CFG), cfm1_audio_sampler.py:107-116 (Euler loop)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from tests.helpers import SEED, clip_batch, describe, exp_noise, gumbel_arrays, gumbel_arrays_steps, rel_l2
from versband_amd import model as vm
from versband_amd import synth

pytestmark = pytest.mark.gpu

LATENT_TOL = 1e-3      # north_star: "within 1e-3 relative mel-latent error"
MEL_L1_TOL = 1e-3      # north_star: "mel L1 vs reference < 1e-3"


@pytest.fixture(scope="module")
def ctx():
    from versband_amd.engine import Context
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return Context("cuda:0")


@pytest.fixture(scope="module")
def prod(ctx):
    """bf16 production engine + VAE decoder on the synthetic checkpoints bench.py uses."""
    from versband_amd.engine import DiTEngine, build_vae_decoder
    dcfg = synth.DiTConfig()
    sd = synth.make_state_dict(synth.dit_shapes(dcfg), SEED)
    sdv = synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), SEED + 1)
    return dict(sd=sd, sdv=sdv, eng=DiTEngine(ctx, dcfg, sd, precision="bf16"), vae=build_vae_decoder(ctx, sdv))


def _oracle_trajectory(sd, inp, scale, steps, noise_steps):
    """ref_cpu.sample_cfg unrolled so that every state x_k and every evaluation's routing indices are kept."""
    T = inp["x_latent"].shape[-1]
    cc = ref_cpu.dit_precompute(sd, inp["t5_cond"], inp["midi"], inp["beats"], T)
    cu = ref_cpu.dit_precompute(sd, inp["t5_uncond"], inp["midi"], inp["beats"], T)
    t_span, idx = ref_cpu.t_index_table(steps + 1)
    x = inp["x_latent"].clone()
    B = x.shape[0]
    states, routes = [x.clone()], []
    t = t_span[0]
    for k in range(steps):
        dt = t_span[k + 1] - t                     # torchdyn's fixed-step bookkeeping, as ref_cpu.sample_cfg restates it
        ti = torch.full((B,), int(idx[k]), dtype=torch.long)
        e_c, aux_c = ref_cpu.dit_forward(sd, x, ti, cc, noise_steps[k][0], return_aux=True)
        e_u, aux_u = ref_cpu.dit_forward(sd, x, ti, cu, noise_steps[k][1], return_aux=True)
        x = x + dt * (e_u + scale * (e_c - e_u))
        t = t + dt
        states.append(x.clone())
        routes.append((aux_c, aux_u))
    return states, routes, [int(v) for v in idx[:steps]]


def _flip_report(eng, cond, states, routes, idx, noise_steps, B, T, depth=4):
    """teacher-forced: the HIP engine evaluates the ORACLE's state x_k, so a differing routing index is a flip caused by
    the arithmetic of this evaluation alone (not by an earlier divergence).  -> flips[k][block] over both gates, both branches."""
    N = B * T
    out = np.zeros((len(idx), depth), dtype=np.int64)
    for k, ti in enumerate(idx):
        t_idx = torch.full((2 * B,), ti, dtype=torch.long)
        _, r = eng.forward(states[k], t_idx, cond, noise=gumbel_arrays(noise_steps[k]), return_routes=True)
        r = r.cpu().long()
        for br in (0, 1):
            aux = routes[k][br]
            for i in range(depth):
                out[k, i] += int((r[i, 0, br * N:(br + 1) * N] != aux[f"ic{i}"]).sum())
                out[k, i] += int((r[i, 1, br * N:(br + 1) * N] != aux[f"ia{i}"]).sum())
    return out


@pytest.mark.parametrize("B,steps", [(2, 10), (1, 50)])
def test_c2_bf16_production_vs_oracle(prod, B, steps):
    """configs[1]'s precision and geometry: bf16 DiT, T = 752, L = 80, CFG scale 3, injected router noise identical to the
    oracle's; 10 steps at B = 2, then the full 50-step schedule for one clip."""
    T, Lc, E, scale = 752, 80, 4, 3.0
    eng, vae, sd, sdv = prod["eng"], prod["vae"], prod["sd"], prod["sdv"]
    inp = clip_batch(B, T, Lc)
    noise_steps = [[exp_noise(B, T, E, 2 * k + br, 4) for br in (0, 1)] for k in range(steps)]
    cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"], inp["beats"], T)
    idx, dts = vm.euler_tables(steps + 1)
    z, traj = eng.sample_cfg(inp["x_latent"], cond, idx, dts, scale, noise=gumbel_arrays_steps(noise_steps), return_traj=True)
    mel = vae.run(z)
    torch.cuda.synchronize()
    states, routes, idx_ref = _oracle_trajectory(sd, inp, scale, steps, noise_steps)
    assert list(idx) == idx_ref
    z_ref = states[-1]
    mel_ref = ref_cpu.vae_decode(sdv, z_ref)
    flips = _flip_report(eng, cond, states, routes, idx_ref, noise_steps, B, T)
    per_step = [rel_l2(traj[k + 1], states[k + 1]) for k in range(steps)]
    decisions = 2 * 2 * B * T                      # per block and evaluation pair: 2 gates x 2 branches x tokens
    print(f"\n[bf16 production, B={B}, {steps} steps] latent rel_l2 = {rel_l2(z, z_ref):.3e}; "
          f"mel L1 = {float((mel.cpu() - mel_ref).abs().mean()):.3e}")
    print("  per-step latent rel_l2: " + " ".join(f"{e:.1e}" for e in per_step))
    print(f"  routing flips per block (teacher-forced, summed over steps, of {decisions * steps} decisions each): "
          f"{flips.sum(0).tolist()}; worst step {int(flips.sum(1).max())}")
    assert torch.isfinite(z).all()
    assert rel_l2(z, z_ref) <= LATENT_TOL, describe(f"bf16 {steps}-step latent vs oracle", z, z_ref)
    assert float((mel.cpu() - mel_ref).abs().mean()) < MEL_L1_TOL, describe("bf16 mel vs oracle", mel, mel_ref)
    # hard routing: flips can only come from near-ties of the two best gate values; they must stay a vanishing fraction
    assert flips.sum() <= 2e-4 * decisions * steps * 4, f"routing flip rate too high: {flips.sum(0).tolist()}"


def _against_fixture(z, mel, name):
    """the oracle's replay of bench.py's clip 0 (tests/golden/<name>, oracle/gen_bench_digest.py): latent in full, mel by digest"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    z_ref = torch.from_numpy(g["z"]).double()
    z = z.detach().double().cpu()
    assert tuple(z.shape) == tuple(z_ref.shape)
    rel = float((z - z_ref).norm() / z_ref.norm())
    m = mel.detach().double().cpu().reshape(-1)
    l1 = float((m[torch.from_numpy(g["mel_idx"])] - torch.from_numpy(g["mel_val"])).abs().mean())
    print(f"{name}: latent rel-L2 {rel:.3e}, mel L1 (sampled) {l1:.3e}")
    assert rel <= LATENT_TOL and l1 < MEL_L1_TOL, f"{name}: latent rel-L2 {rel:.3e}, mel L1 {l1:.3e}"


def test_c3_e8_bf16_production_clip_vs_oracle_fixture(ctx, prod):
    """BASELINE configs[2] in the benchmarked precision: clip 0 of `bench.py --workload c3` (8 experts per group, router noise drawn
    on the device) over all 50 CFG Euler steps + VAE decode against the oracle's replay; it rides in a batch of 3 here (a clip's
    result does not depend on its batch)."""
    from versband_amd.engine import DiTEngine
    dcfg = synth.DiTConfig(num_experts=8)
    eng = DiTEngine(ctx, dcfg, synth.make_state_dict(synth.dit_shapes(dcfg), SEED), precision="bf16")
    T, Lc, B = 752, 80, 3
    inp = clip_batch(B, T, Lc, clip0=0, seed=SEED)
    idx, dts = vm.euler_tables(51)
    cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"], inp["beats"], T)
    z = eng.sample_cfg(inp["x_latent"], cond, idx, dts, 3.0, seed=SEED, clip_base=0)
    mel = prod["vae"].run(z[:1].contiguous())
    torch.cuda.synchronize()
    _against_fixture(z[:1], mel, "bench_clip0_e8.npz")


def test_c5_longform_bf16_production_clip_vs_oracle_fixture(prod):
    """BASELINE configs[4] in the benchmarked precision: clip 0 of `bench.py --workload c5` (4 x 120 s, T = 4500: four windows of 1500
    as batch rows, linear cross-fade, VAE decode of the whole latent) against the oracle's replay of its four windows."""
    from versband_amd import longform
    T, Lc, B = 4500, 80, 4
    inp = clip_batch(B, T, Lc, clip0=0, seed=SEED)
    idx, dts = vm.euler_tables(51)
    dev = "cuda:0"
    z = longform.sample_long(prod["eng"], inp["x_latent"].to(dev), inp["t5_cond"].to(dev), inp["t5_uncond"].to(dev), inp["midi"].to(dev),
                             inp["beats"].to(dev), idx, dts, 3.0, window=1500, overlap=128, seed=SEED, clip_base=0)
    mel = prod["vae"].run(z[:1].contiguous())
    torch.cuda.synchronize()
    _against_fixture(z[:1], mel, "bench_clip0_long.npz")


def test_c2_second_stream_clip_and_replay_seed_vs_oracle_fixture(prod):
    """what bench.py verifies beyond clip 0 of pass 0 (tests/golden/bench_c2_checks.npz): global clip 4 - the first clip of the
    second sub-batch / stream - and global clip 7 - the LAST row of that sub-batch, the partial-tile edge of every token-row launch
    (round 4) - with the sampler seeds of passes 0 and 1, riding in a batch of 4 at clip_base 4 as they do in the benchmark."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bench_c2_checks.npz"))
    assert [int(c) for c in g["clips"]] == [0, 4, 7]
    T, Lc, B = 752, 80, 4
    inp = clip_batch(B, T, Lc, clip0=4, seed=SEED)
    idx, dts = vm.euler_tables(51)
    eng = prod["eng"]
    cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"], inp["beats"], T)
    for ps in (0, 1):
        z = eng.sample_cfg(inp["x_latent"], cond, idx, dts, 3.0, seed=SEED + ps, clip_base=4)
        for clip in (4, 7):
            r = clip - 4
            mel = prod["vae"].run(z[r:r + 1].contiguous())
            torch.cuda.synchronize()
            z_ref = torch.from_numpy(g[f"z_c{clip}_p{ps}"]).double()
            rel = float((z[r:r + 1].double().cpu() - z_ref).norm() / z_ref.norm())
            m = mel.double().cpu().reshape(-1)
            l1 = float((m[torch.from_numpy(g[f"mel_c{clip}_p{ps}_idx"])] - torch.from_numpy(g[f"mel_c{clip}_p{ps}_val"])).abs().mean())
            print(f"clip {clip}, pass {ps}: latent rel-L2 {rel:.3e}, mel L1 {l1:.3e}")
            assert rel <= LATENT_TOL and l1 < MEL_L1_TOL


@pytest.mark.parametrize("B,vprec", [(1, "fp32"), (2, "fp32"), (1, "fp32mf"), (2, "fp32mf")])
def test_path_bits_are_stable_beside_a_second_gpu_process(ctx, prod, B, vprec):
    """Every counted wait of the DMA rings has to hold when the memory system is busy with someone else's traffic - the condition under
    which round 5 found a wait one piece short in the fp32 conv (profiles/r05_conv_tail_race.txt; two ranks sharing a GPU are exactly
    this).  One or two clips - the sizes whose tile choices differ from the 8-clip bench - through sampler (3 Euler steps), fp32 VAE
    decoder and fp32 vocoder, repeated beside a process that streams HBM: latents, mels and waveforms have the first run's bits."""
    from tests.helpers import beside_load
    from versband_amd.engine import build_hifigan, build_vae_decoder
    hcfg = synth.HifiGanConfig()
    vae = build_vae_decoder(ctx, prod["sdv"], precision=vprec)
    voc = build_hifigan(ctx, synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2), hcfg.as_hparams(), precision=vprec)
    T, Lc = 752, 80
    inp = clip_batch(B, T, Lc)
    t5 = torch.cat([inp["t5_cond"], inp["t5_uncond"]])
    idx, dts = vm.euler_tables(4)

    def run():
        cond = prod["eng"].precompute_cond(t5, inp["midi"], inp["beats"], T)
        z = prod["eng"].sample_cfg(inp["x_latent"], cond, idx, dts, 3.0, seed=11)
        mel = vae.run(z)
        wav = voc.run(mel)
        torch.cuda.synchronize()
        return z.clone(), mel.clone(), wav.clone()

    with beside_load(60) as load:
        first = run()
        assert all(torch.isfinite(t).all() for t in first)
        for rep in range(12):
            cur = run()
            for name, a, b in zip(("latent", "mel", "waveform"), cur, first):
                assert torch.equal(a, b), f"run {rep + 1}: {name} differs from the first run by {float((a - b).abs().max()):.3e}"
        assert load.alive(), "the load process ended before the repeats did"


def _repeat_beside_load(run, names, repeats, seconds):
    """run() -> tuple of tensors; first run + `repeats` more beside the load process, every tensor with the first run's bits"""
    from tests.helpers import beside_load
    with beside_load(seconds) as load:
        first = run()
        assert all(torch.isfinite(t).all() for t in first)
        for rep in range(repeats):
            for name, a, b in zip(names, run(), first):
                assert torch.equal(a, b), f"run {rep + 1}: {name} differs from the first run by {float((a - b).abs().max()):.3e}"
        assert load.alive(), "the load process ended before the repeats did"


@pytest.mark.parametrize("vprec", ["fp32", "fp32mf"])
def test_vocoder_stage_bits_are_stable_beside_a_second_gpu_process_at_8_clips(ctx, prod, vprec):
    """Round 5's race lived in a path the suite did not stress (verdict item 7): the bench's own vocoder-heavy shape - fp32 VAE decoder
    + fp32 HiFi-GAN over 8 clips of 20 s (the tile choices of the 8-clip launches: 128 x 96 VAE tiles, fused ResBlock pairs, one-round
    1 x 1 layers; with "fp32mf" the minimal-filtering kernel's rings on both of its tiles) - repeated beside the load process, bitwise."""
    from versband_amd.engine import build_hifigan, build_vae_decoder
    hcfg = synth.HifiGanConfig()
    vae = build_vae_decoder(ctx, prod["sdv"], precision=vprec)
    voc = build_hifigan(ctx, synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2), hcfg.as_hparams(), precision=vprec)
    z = clip_batch(8, 752, 80)["x_latent"].to("cuda:0")

    def run():
        mel = vae.run(z)
        wav = voc.run(mel)
        torch.cuda.synchronize()
        return mel.clone(), wav.clone()

    _repeat_beside_load(run, ("mel", "waveform"), 6, 60)


def test_e8_dit_bits_are_stable_beside_a_second_gpu_process(ctx):
    """configs[2]'s DiT (8 experts per group: 64 pair buckets, the 96-column band kernel, grouped SwiGLU over 16 groups) at 8 clips, three
    CFG Euler steps through the sampler, repeated beside the load process: latents bitwise equal."""
    from versband_amd.engine import DiTEngine
    dcfg = synth.DiTConfig(num_experts=8)
    eng = DiTEngine(ctx, dcfg, synth.make_state_dict(synth.dit_shapes(dcfg), SEED), precision="bf16")
    B, T, Lc = 8, 752, 80
    inp = clip_batch(B, T, Lc)
    t5 = torch.cat([inp["t5_cond"], inp["t5_uncond"]])
    idx, dts = vm.euler_tables(4)

    def run():
        cond = eng.precompute_cond(t5, inp["midi"], inp["beats"], T)
        z = eng.sample_cfg(inp["x_latent"], cond, idx, dts, 3.0, seed=5)
        torch.cuda.synchronize()
        return (z.clone(),)

    _repeat_beside_load(run, ("latent",), 10, 60)


def test_longform_bits_are_stable_beside_a_second_gpu_process(ctx, prod):
    """configs[4]'s path: two 120 s clips = 8 windows of 1500 tokens in one sampler batch (three Euler steps), the cross-fade kernel, the
    fp32 VAE over 9000 mel frames and the halo'd chunked fp32 vocoder, repeated beside the load process: every stage bitwise equal."""
    from versband_amd import longform
    from versband_amd.engine import build_hifigan, build_vae_decoder
    hcfg = synth.HifiGanConfig()
    vae = build_vae_decoder(ctx, prod["sdv"], precision="fp32")
    voc = build_hifigan(ctx, synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2), hcfg.as_hparams(), precision="fp32")
    B, T, Lc = 2, 4500, 80
    dev = "cuda:0"
    inp = {k: v.to(dev) for k, v in clip_batch(B, T, Lc).items()}
    idx, dts = vm.euler_tables(4)

    def run():
        z = longform.sample_long(prod["eng"], inp["x_latent"], inp["t5_cond"], inp["t5_uncond"], inp["midi"], inp["beats"], idx, dts, 3.0,
                                 window=1500, overlap=128, seed=9, clip_base=0)
        mel = vae.run(z)
        wav = longform.vocode_chunked(voc, mel, chunk=3000, halo=32)
        torch.cuda.synchronize()
        return z.clone(), mel.clone(), wav.clone()

    _repeat_beside_load(run, ("latent", "mel", "waveform"), 3, 90)
