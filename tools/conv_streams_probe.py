"""VAE decoder + vocoder of 8 clips as 1 x 8, 2 x 4, 4 x 2 and 8 x 1 concurrent sub-batches (one HIP stream + host thread each): wall time of
the whole set.  The 2-stream pass runs 2 x 4 today.      python tools/conv_streams_probe.py [fp32|split]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, build_hifigan, build_vae_decoder  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
device = torch.device("cuda:0")
vcfg, hcfg = synth.VAEConfig(), synth.HifiGanConfig()
sds = [synth.make_state_dict(s, 1235 + i) for i, s in enumerate([synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)])]
ctx = Context(device)
TOTAL, T = 8, 752
z_all = torch.randn(TOTAL, 20, T, device=device)


def make(n):
    return dict(vae=build_vae_decoder(ctx, sds[0], precision=prec), voc=build_hifigan(ctx, sds[1], hcfg.as_hparams(), precision=prec),
                stream=torch.cuda.Stream(device=device), n=n)


def run(w, z, reps):
    torch.cuda.set_device(device)
    with torch.cuda.stream(w["stream"]):
        for _ in range(reps):
            w["wav"] = w["voc"].run(w["vae"].run(z))


ref = None
for S in (1, 2, 4, 8):
    n = TOTAL // S
    ws = [make(n) for _ in range(S)]
    zs = [z_all[i * n:(i + 1) * n].contiguous() for i in range(S)]
    for w, z in zip(ws, zs):
        run(w, z, 2)
    torch.cuda.synchronize()
    wav = torch.cat([w["wav"] for w in ws])
    if ref is None:
        ref = wav.clone()
    same = torch.equal(wav, ref)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        ths = [threading.Thread(target=run, args=(w, z, 2)) for w, z in zip(ws, zs)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 2)
    print(f"{S} x {n} clips: {best * 1e3:7.2f} ms per 8 clips   (waveforms equal to the 1 x 8 run: {same})", flush=True)
    del ws
    torch.cuda.empty_cache()
