"""Find the first op of the VAE decoder program whose result is not run-to-run deterministic (B = 2, beside a load process)."""
import copy
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import _lib as L  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, ConvNet, build_hifigan, build_vae_decoder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
which = sys.argv[2] if len(sys.argv) > 2 else "vae"
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
    ctx = Context(device)
    if which == "vae":
        sd = synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), 1235)
        net = build_vae_decoder(ctx, sd, precision="fp32")
        x = torch.randn(B, 20, 752, device=device)
    else:
        hcfg = synth.HifiGanConfig()
        sd = synth.make_state_dict(synth.hifigan_shapes(hcfg), 1236)
        net = build_hifigan(ctx, sd, hcfg.as_hparams(), precision="fp32")
        x = torch.randn(B, 80, 1504, device=device) * 0.5
    nb = net.nb
    nops = len(nb.ops)
    print("ops", nops, "bufs", len(nb.bufs), flush=True)
    KN = {v: k for k, v in vars(L).items() if k.startswith("OP_")}

    def differs(k, reps=6):
        nb2 = copy.copy(nb)
        nb2.ops = list(nb.ops[:k])
        sub = ConvNet(ctx, net.which, nb2, net.in_ch, net.out_ch, net.out_tmul, net.in_tmul)
        ws = sub._workspace(B, x.shape[2] // net.in_tmul)
        ws.zero_()
        sub.run(x)
        torch.cuda.synchronize()
        ref = ws.clone()
        n = 0
        for _ in range(reps):
            ws.zero_()
            sub.run(x)
            torch.cuda.synchronize()
            n += int(not torch.equal(ws, ref))
        return n

    full = differs(nops)
    print("full program: runs whose workspace differs:", full, flush=True)
    lo, hi = 0, nops            # differs(lo) == 0 assumed, differs(hi) > 0
    if full:
        while hi - lo > 1:
            mid = (lo + hi) // 2
            d = differs(mid)
            print(f"  first {mid} ops: {d} of 6 differ", flush=True)
            if d:
                hi = mid
            else:
                lo = mid
        o = nb.ops[hi - 1]
        print(f"first non-deterministic op: index {hi - 1}, kind {KN.get(o.kind, o.kind)}, Ci {o.Ci} Co {o.Co} k {o.ksize} dil {o.dil} x {o.x} out {o.out} res {o.res} "
              f"in_act {o.in_act} upsample2 {getattr(o, 'upsample2', None)} alpha {o.alpha} beta {o.beta}")
        for j in range(max(0, hi - 4), min(nops, hi + 2)):
            oo = nb.ops[j]
            print(f"    op {j}: kind {KN.get(oo.kind, oo.kind)} Ci {oo.Ci} Co {oo.Co} k {oo.ksize} x {oo.x} out {oo.out} res {oo.res} stats {oo.stats}")
finally:
    child.kill()
