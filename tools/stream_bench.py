"""Experiment: does running S independent sub-batches on S HIP streams (S host threads) beat one batch of 8?"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import model as vm  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, DiTEngine, build_hifigan, build_vae_decoder  # noqa: E402

SEED, T, Lc = 1234, 752, 80
dev = torch.device("cuda:0")
dcfg, vcfg, hcfg = synth.DiTConfig(), synth.VAEConfig(), synth.HifiGanConfig()
sds = [synth.make_state_dict(s, SEED + i) for i, s in enumerate([synth.dit_shapes(dcfg), synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)])]
ctx = Context(dev)
idx, dts = vm.euler_tables(51)
for S in (1, 2, 4):
    B = 8 // S
    workers = []
    for s in range(S):
        eng = DiTEngine(ctx, dcfg, sds[0], precision="bf16")
        vae = build_vae_decoder(ctx, sds[1])
        voc = build_hifigan(ctx, sds[2], hcfg.as_hparams())
        inp = clip_batch(B, T, Lc, clip0=s * B, seed=SEED)
        x0 = inp["x_latent"].to(dev)
        t5 = torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(dev)
        midi, beats = inp["midi"].to(dev), inp["beats"].to(dev)
        workers.append((eng, vae, voc, x0, t5, midi, beats, torch.cuda.Stream(), s * B))

    def run(w, n):
        eng, vae, voc, x0, t5, midi, beats, st, cb = w
        with torch.cuda.stream(st):
            for k in range(n):
                cond = eng.precompute_cond(t5, midi, beats, T)
                z = eng.sample_cfg(x0, cond, idx, dts, 3.0, seed=SEED + k, clip_base=cb)
                voc.run(vae.run(z))

    def all_run(n):
        ths = [threading.Thread(target=run, args=(w, n)) for w in workers]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
    all_run(1)
    t0 = time.perf_counter()
    all_run(2)
    el = time.perf_counter() - t0
    print(f"streams={S} sub-batch={B}: {el / 2 * 1e3:.1f} ms per 8 clips -> {8 * 20.0 * 2 / el:.0f} mel-s/s", flush=True)
    del workers
    torch.cuda.empty_cache()
