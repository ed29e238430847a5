"""How well do the DiT sampler and the VAE + vocoder convolutions share the GPU?  Times, for sub-batches of `n` clips:
  (a) the sampler alone, (b) VAE + vocoder alone, (c) both at once on two streams (different clips), (d) two samplers at once, (e) two
  VAE + vocoder runs at once - what today's 2-stream pass does.  If (c) is well under (a) + (b) a pass that pairs one sub-batch's sampler
  with the other's convolutions beats one that runs like with like.     python tools/overlap_probe.py [clips per sub-batch = 4] [euler steps = 10]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
device = torch.device("cuda:0")
T, Lc = 752, 80
dcfg, vcfg, hcfg = synth.DiTConfig(), synth.VAEConfig(), synth.HifiGanConfig()
sds = [synth.make_state_dict(s, 1234 + i) for i, s in enumerate([synth.dit_shapes(dcfg), synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)])]
ctx = Context(device)
idx, dts = vm.euler_tables(steps + 1)


def worker(clip0, share=None):
    eng = DiTEngine(ctx, dcfg, sds[0], precision="bf16", share=share)
    inp = clip_batch(n, T, Lc, clip0=clip0)
    w = dict(eng=eng, vae=build_vae_decoder(ctx, sds[1], precision=prec), voc=build_hifigan(ctx, sds[2], hcfg.as_hparams(), precision=prec),
             x0=inp["x_latent"].to(device), t5=torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(device), midi=inp["midi"].to(device),
             beats=inp["beats"].to(device), stream=torch.cuda.Stream(device=device))
    return w


def dit(w, reps):
    for k in range(reps):
        cond = w["eng"].precompute_cond(w["t5"], w["midi"], w["beats"], T, persistent=True)
        w["z"] = w["eng"].sample_cfg(w["x0"], cond, idx, dts, 3.0, seed=7 + k)


def conv(w, reps):
    for _ in range(reps):
        w["wav"] = w["voc"].run(w["vae"].run(w["z"]))


A = worker(0)
Bw = worker(n, share=A["eng"])
for w in (A, Bw):
    with torch.cuda.stream(w["stream"]):
        dit(w, 3)          # (the third call replays the captured graph)
        conv(w, 2)
torch.cuda.synchronize()


def timed(jobs, reps=3):
    """jobs: list of (fn, worker, inner reps); each on its worker's stream from its own host thread; best wall time of `reps`"""
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()

        def run(fn, w, r):
            torch.cuda.set_device(device)
            with torch.cuda.stream(w["stream"]):
                fn(w, r)
        ths = [threading.Thread(target=run, args=j) for j in jobs]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


R = 2
a = timed([(dit, A, R)])
b = timed([(conv, Bw, R)])
# pick inner repeat counts so that both sides of (c) take about equally long alone
rb = max(1, round(R * a / b))
b2 = timed([(conv, Bw, rb)])
c = timed([(dit, A, R), (conv, Bw, rb)])
d = timed([(dit, A, R), (dit, Bw, R)])
e = timed([(conv, A, rb), (conv, Bw, rb)])
print(f"{n} clips per sub-batch, {steps} Euler steps x 2 NFE per sampler call, VAE / vocoder {prec}")
print(f"(a) sampler alone x{R}:                   {a:8.2f} ms")
print(f"(b) VAE + vocoder alone x{rb}:             {b2:8.2f} ms   (x{R}: {b:.2f})")
print(f"(c) sampler || VAE + vocoder:              {c:8.2f} ms   = {c / (a + b2):.3f} of (a) + (b)")
print(f"(d) sampler || sampler:                    {d:8.2f} ms   = {d / (2 * a):.3f} of 2 (a)")
print(f"(e) VAE + vocoder || VAE + vocoder:        {e:8.2f} ms   = {e / (2 * b2):.3f} of 2 (b)")
print(f"like with like: (d) + (e) = {d + e:.2f} ms; crossed: 2 (c) = {2 * c:.2f} ms  -> {100 * (1 - 2 * c / (d + e)):+.1f} % time")
