"""Experiment: sub-batch split (2 full pipelines on 2 streams) vs stage pipeline (DiT of step k+1 overlapping VAE+vocoder of step k)."""
import os
import queue
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import model as vm  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, DiTEngine, build_hifigan, build_vae_decoder  # noqa: E402

SEED, T, Lc, K = 1234, 752, 80, int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
dcfg, vcfg, hcfg = synth.DiTConfig(), synth.VAEConfig(), synth.HifiGanConfig()
sds = [synth.make_state_dict(s, SEED + i) for i, s in enumerate([synth.dit_shapes(dcfg), synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)])]
ctx = Context(dev)
idx, dts = vm.euler_tables(51)


def mk(B, clip0):
    inp = clip_batch(B, T, Lc, clip0=clip0, seed=SEED)
    return dict(eng=DiTEngine(ctx, dcfg, sds[0], precision="bf16"), vae=build_vae_decoder(ctx, sds[1]), voc=build_hifigan(ctx, sds[2], hcfg.as_hparams()),
                x0=inp["x_latent"].to(dev), t5=torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(dev), midi=inp["midi"].to(dev),
                beats=inp["beats"].to(dev), cb=clip0)


def dit(w, k):
    cond = w["eng"].precompute_cond(w["t5"], w["midi"], w["beats"], T)
    return w["eng"].sample_cfg(w["x0"], cond, idx, dts, 3.0, seed=SEED + k, clip_base=w["cb"])


def split_mode(n):
    ws = [mk(4, 0), mk(4, 4)]
    sts = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(w, st, n):
        with torch.cuda.stream(st):
            for k in range(n):
                w["voc"].run(w["vae"].run(dit(w, k)))

    def go(n):
        ths = [threading.Thread(target=run, args=(w, st, n)) for w, st in zip(ws, sts)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        torch.cuda.synchronize()
    go(1)
    t0 = time.perf_counter()
    go(n)
    return (time.perf_counter() - t0) / n


def pipe_mode(n, nsub):
    ws = [mk(8 // nsub, i * (8 // nsub)) for i in range(nsub)]
    sd, sv = torch.cuda.Stream(), torch.cuda.Stream()

    def go(n):
        q = queue.Queue()

        def producer():
            with torch.cuda.stream(sd):
                for k in range(n):
                    for w in ws:
                        q.put((w, dit(w, k)))        # sample_cfg returns after its stream is idle: z is ready
            q.put(None)

        def consumer():
            with torch.cuda.stream(sv):
                while True:
                    it = q.get()
                    if it is None:
                        break
                    w, z = it
                    w["voc"].run(w["vae"].run(z))
        ths = [threading.Thread(target=producer), threading.Thread(target=consumer)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        torch.cuda.synchronize()
    go(1)
    t0 = time.perf_counter()
    go(n)
    return (time.perf_counter() - t0) / n


for name, fn in (("split 2x4", lambda: split_mode(K)), ("pipeline 1x8", lambda: pipe_mode(K, 1)), ("pipeline 2x4", lambda: pipe_mode(K, 2)),
                 ("split 2x4", lambda: split_mode(K))):
    dt = fn()
    print(f"{name}: {dt * 1e3:.1f} ms per 8 clips -> {160.0 / dt:.0f} mel-s/s ({K} steps)", flush=True)
    torch.cuda.empty_cache()
