"""Shader clock / board power PER PHASE of the c2 pass (run on the GPU box): the sampler alone, the fp32 VAE + vocoder alone, the
bf16x3 VAE + vocoder alone - each looped for a few seconds with bench.py's sysfs sampler beside it.  The f32-MFMA peak quoted in the
roofline (157.3 TF) assumes 2.4 GHz; what a phase actually ran at belongs next to its fraction.

    python tools/phase_clocks.py [seconds per phase, default 4] [batch, default 8]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import L_CTX, SEED, Telemetry  # noqa: E402
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import model as vm  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, DiTEngine, build_hifigan, build_vae_decoder  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
device = torch.device("cuda:0")
T_lat = 752
dcfg, vcfg, hcfg = synth.DiTConfig(), synth.VAEConfig(), synth.HifiGanConfig()
sds = [synth.make_state_dict(s, SEED + i) for i, s in enumerate([synth.dit_shapes(dcfg), synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)])]
ctx = Context(device)
eng = DiTEngine(ctx, dcfg, sds[0], precision="bf16")
inp = clip_batch(B, T_lat, L_CTX, clip0=0, seed=SEED)
x0 = inp["x_latent"].to(device)
t5 = torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(device)
midi, beats = inp["midi"].to(device), inp["beats"].to(device)
idx, dts = vm.euler_tables(51)
cond = eng.precompute_cond(t5, midi, beats, T_lat, persistent=True)
z = eng.sample_cfg(x0, cond, idx, dts, 3.0, seed=SEED, clip_base=0)
nets = {p: (build_vae_decoder(ctx, sds[1], precision=p), build_hifigan(ctx, sds[2], hcfg.as_hparams(), precision=p)) for p in ("fp32", "split")}


def phase(name, fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    with Telemetry(period=0.01) as tel:
        while time.perf_counter() - t0 < secs:
            fn()
            torch.cuda.synchronize()
            n += 1
        ms = 1e3 * (time.perf_counter() - t0) / n
    s = tel.summary(device)
    print(f"{name:28s} {ms:8.2f} ms per run  x{n:3d}   sclk {s.get('sclk_mhz_min')}..{s.get('sclk_mhz_max')} avg {s.get('sclk_mhz_avg')} MHz, "
          f"{s.get('power_w_avg')} W of {s.get('power_cap_w')} ({s.get('samples')} samples)", flush=True)


phase("sampler (50 steps, bf16)", lambda: eng.sample_cfg(x0, cond, idx, dts, 3.0, seed=SEED, clip_base=0))
for p in ("fp32", "split"):
    vae, voc = nets[p]
    mel = vae.run(z)
    phase(f"VAE decoder ({p})", lambda: vae.run(z))
    phase(f"vocoder ({p})", lambda: voc.run(mel))
