"""Does a clip's result depend on the batch it rides in?  Stage by stage, clips [0,1] alone vs the same clips inside a batch of 4
(and 8): DiT forward, 3-step sampler, VAE decode, HiFi-GAN.  Bitwise comparison; prints max |d| per stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.helpers import SEED, clip_batch
from versband_amd import model as vm, synth
from versband_amd.engine import Context, DiTEngine, build_hifigan, build_vae_decoder

ctx = Context("cuda:0")
dcfg, vcfg, hcfg = synth.DiTConfig(), synth.VAEConfig(), synth.HifiGanConfig()
sd = synth.make_state_dict(synth.dit_shapes(dcfg), SEED)
sdv = synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1)
sdh = synth.make_state_dict(synth.hifigan_shapes(hcfg), SEED + 2)
eng = DiTEngine(ctx, dcfg, sd, precision="bf16")
vae = build_vae_decoder(ctx, sdv)
voc = build_hifigan(ctx, sdh, hcfg.as_hparams())
T, Lc = 752, 80
idx, dts = vm.euler_tables(4)


def stage_outputs(B):
    inp = clip_batch(B, T, Lc)
    cond = eng.precompute_cond(torch.cat([inp["t5_cond"], inp["t5_uncond"]]), inp["midi"], inp["beats"], T)
    t_idx = torch.full((2 * B,), 321, dtype=torch.long)
    v, r = eng.forward(inp["x_latent"], t_idx, cond, seed=7, clip_base=0, return_routes=True)
    v = v.clone(); r = r.clone()
    z = eng.sample_cfg(inp["x_latent"], cond, idx, dts, 3.0, seed=7, clip_base=0).clone()
    zfix = torch.from_numpy(synth.prng.normal(91, 8 * 20 * T).reshape(8, 20, T)).float()[:B].cuda()
    mel = vae.run(zfix).clone()
    melfix = torch.from_numpy(synth.prng.uniform(92, 8 * 80 * 2 * T, -5.0, 1.0).reshape(8, 80, 2 * T)).float()[:B].cuda()
    wav = voc.run(melfix).clone()
    torch.cuda.synchronize()
    N = B * T
    return dict(v_cond=v[:2].cpu(), v_unc=v[B:B + 2].cpu(), routes=torch.cat([r[:, :, :2 * T], r[:, :, N:N + 2 * T]], 2).cpu(),
                z=z[:2].cpu(), mel=mel[:2].cpu(), wav=wav[:2].cpu())


ref = stage_outputs(2)
for B in (4, 8):
    got = stage_outputs(B)
    for k in ref:
        a, b = ref[k].double(), got[k].double()
        print(f"B=2 vs B={B}  {k:8s} equal={torch.equal(ref[k], got[k])}  max|d|={float((a - b).abs().max()):.3e}  ndiff={int((a != b).sum())}", flush=True)
