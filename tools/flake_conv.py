"""Which launch parameters of the DMA-fed fp32 conv make the VAE's first ops non-deterministic beside a second GPU process?
(experiments build: VB_F32G_PICK / VB_F32G_LDSPAD / VB_F32G_NOSTAGE / VB_F32G_OLDWAIT are read at every launch)   python tools/flake_conv.py [ops] [reps]"""
import copy
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, ConvNet, build_vae_decoder  # noqa: E402

B = 2
K = int(sys.argv[1]) if len(sys.argv) > 1 else 11
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if os.environ.get("FLAKE_LOAD_CHILD"):
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(8, 256, 60000, device="cuda")
    while True:
        for _ in range(20):
            a @ a
            torch.nn.functional.leaky_relu(x, 0.1)
        torch.cuda.synchronize()

CONFIGS = [
    ("pick2 (64x128, 4-stage ring, 40 KB)", dict(VB_F32G_PICK="2")),
    ("pick2, round-4 wait count", dict(VB_F32G_PICK="2", VB_F32G_OLDWAIT="1")),
    ("pick2, round-4 wait count, 3 wg/CU", dict(VB_F32G_PICK="2", VB_F32G_OLDWAIT="1", VB_F32G_LDSPAD="8192")),
    ("pick5 (32x256), round-4 wait count", dict(VB_F32G_PICK="5", VB_F32G_OLDWAIT="1")),
    ("default tile choice, round-4 wait count", dict(VB_F32G_OLDWAIT="1")),
    ("default tile choice", dict()),
    ("pick4 (64x128, 3-stage ring, 36 KB)", dict(VB_F32G_PICK="4")),
    ("pick2 + 8 KB unused LDS (3 wg/CU)", dict(VB_F32G_PICK="2", VB_F32G_LDSPAD="8192")),
    ("pick2 + 24 KB unused LDS (2 wg/CU)", dict(VB_F32G_PICK="2", VB_F32G_LDSPAD="24576")),
    ("pick2, plain epilogue", dict(VB_F32G_PICK="2", VB_F32G_NOSTAGE="1")),
    ("pick5 (32x256, 4-stage ring)", dict(VB_F32G_PICK="5")),
    ("pick3 (64x64)", dict(VB_F32G_PICK="3")),
    ("pick1 (128x96)", dict(VB_F32G_PICK="1")),
    ("pick0 (128x128)", dict(VB_F32G_PICK="0")),
]
KNOBS = ("VB_F32G_PICK", "VB_F32G_LDSPAD", "VB_F32G_NOSTAGE", "VB_F32G_OLDWAIT")


def main():
    load = os.environ.get("FLAKE_NO_LOAD") is None
    child = subprocess.Popen([sys.executable, __file__], env=dict(os.environ, FLAKE_LOAD_CHILD="1")) if load else None
    try:
        device = torch.device("cuda:0")
        ctx = Context(device)
        sd = synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), 1235)
        net = build_vae_decoder(ctx, sd, precision="fp32")
        x = torch.randn(B, 20, 752, device=device)
        nb = net.nb
        nb2 = copy.copy(nb)
        nb2.ops = list(nb.ops[:K])
        sub = ConvNet(ctx, net.which, nb2, net.in_ch, net.out_ch, net.out_tmul, net.in_tmul)
        T = 752
        ws = sub._workspace(B, T)

        def run():
            ws.zero_()
            sub.run(x)
            torch.cuda.synchronize()
            return ws.clone()

        print(f"first {K} ops of the VAE decoder, {B} clips, {REPS} runs per configuration, second process: {'yes' if load else 'no'}", flush=True)
        base = None
        for name, env in CONFIGS:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            ref = run()
            bad = sum(int(not torch.equal(run(), ref)) for _ in range(REPS))
            same_as_first = "" if base is None else ("  (= first configuration's bytes)" if torch.equal(ref, base) else "  (reference run differs from the first configuration's)")
            if base is None:
                base = ref
            print(f"  {name:42s}: {bad:3d} of {REPS} runs differ from the configuration's first run{same_as_first}", flush=True)
    finally:
        if child:
            child.kill()


main()
