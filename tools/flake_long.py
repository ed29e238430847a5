"""Run-to-run determinism of the long-form path (windows of 1500 tokens, crossfades, chunked vocoder) beside a second GPU process.
    python tools/flake_long.py [clips=1] [latent frames=3000] [reps=5] [fp32|split]"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import longform  # noqa: E402
from versband_amd import model as vm  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, DiTEngine, build_hifigan, build_vae_decoder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
PREC = sys.argv[4] if len(sys.argv) > 4 else "fp32"
if os.environ.get("FLAKE_LOAD_CHILD"):
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(8, 256, 60000, device="cuda")
    while True:
        for _ in range(20):
            a @ a
            torch.nn.functional.leaky_relu(x, 0.1)
        torch.cuda.synchronize()
child = subprocess.Popen([sys.executable, __file__], env=dict(os.environ, FLAKE_LOAD_CHILD="1"))
try:
    device = torch.device("cuda:0")
    dcfg, vcfg, hcfg = synth.DiTConfig(), synth.VAEConfig(), synth.HifiGanConfig()
    sds = [synth.make_state_dict(s, 1234 + i) for i, s in enumerate([synth.dit_shapes(dcfg), synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)])]
    ctx = Context(device)
    eng = DiTEngine(ctx, dcfg, sds[0], precision="bf16")
    vae = build_vae_decoder(ctx, sds[1], precision=PREC)
    voc = build_hifigan(ctx, sds[2], hcfg.as_hparams(), precision=PREC)
    inp = clip_batch(B, T, 80)
    idx, dts = vm.euler_tables(4)
    time.sleep(15)

    def run():
        z = longform.sample_long(eng, inp["x_latent"].to(device), inp["t5_cond"].to(device), inp["t5_uncond"].to(device), inp["midi"].to(device),
                                 inp["beats"].to(device), idx, dts, 3.0, window=dcfg.max_len, overlap=128, seed=5)
        mel = vae.run(z)
        wav = longform.vocode_chunked(voc, mel, chunk=3000, halo=32)
        torch.cuda.synchronize()
        return z.clone(), mel.clone(), wav.clone()

    first = run()
    bad = 0
    for rep in range(REPS):
        cur = run()
        diffs = [name for name, a, b in zip(("latent", "mel", "waveform"), cur, first) if not torch.equal(a, b)]
        bad += bool(diffs)
        if diffs:
            print(f"run {rep + 1}: {diffs} differ")
    print(f"long-form {B} clip(s) x {T} latent frames ({T / 37.5:.0f} s), VAE / vocoder {PREC}: {bad} of {REPS} runs differ from the first", flush=True)
finally:
    child.kill()
