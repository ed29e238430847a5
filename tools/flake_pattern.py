"""Where do two runs of the first non-deterministic VAE op differ?  (experiments build, VB_F32G_PICK=2, B=2, beside a load process)"""
import copy
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from versband_amd import synth  # noqa: E402
from versband_amd.engine import Context, ConvNet, build_vae_decoder  # noqa: E402

B, K = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 11
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
    sd = synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), 1235)
    net = build_vae_decoder(ctx, sd, precision="fp32")
    x = torch.randn(B, 20, 752, device=device)
    nb = net.nb
    nb2 = copy.copy(nb)
    nb2.ops = list(nb.ops[:K])
    sub = ConvNet(ctx, net.which, nb2, net.in_ch, net.out_ch, net.out_tmul, net.in_tmul)
    T = 752
    offs, off = [], 0
    for (ch, tmul, sq) in nb.bufs:
        tl = T * tmul
        el = tl * tl if sq == 1 else (ch * ((tl + 31) // 32 * 32) if sq == 2 else (ch * (tl + 64 + 384) if sq == 3 else ch * tl))
        offs.append(off)
        off = (off + el * B * 4 + 255) // 256 * 256
    o = nb.ops[K - 1]
    ch, tmul, sq = nb.bufs[o.out]
    tl = T * tmul
    n = (tl * tl if sq == 1 else ch * tl) * B
    ws = sub._workspace(B, T)

    def run():
        ws.zero_()
        sub.run(x)
        torch.cuda.synchronize()
        return ws[offs[o.out]:offs[o.out] + 4 * n].view(torch.float32).clone()

    ref = run()
    rows = tl if sq == 1 else ch
    for rep in range(8):
        cur = run()
        d = (cur != ref).view(B, rows, tl)
        if d.any():
            idx = d.nonzero()
            bs = sorted(set(idx[:, 0].tolist())); cs = idx[:, 1]; ts = idx[:, 2]
            dv = (cur.view(B, rows, tl) - ref.view(B, rows, tl)).abs()
            print(f"rep {rep}: {int(d.sum())} elements differ; clips {bs}; channel rows {int(cs.min())}..{int(cs.max())} ({len(set(cs.tolist()))} distinct, "
                  f"co tiles {sorted(set((cs // 64).tolist()))[:12]}); positions {int(ts.min())}..{int(ts.max())} ({len(set(ts.tolist()))} distinct, "
                  f"t tiles {sorted(set((ts // 128).tolist()))}); max |d| {float(dv.max()):.3e}; nan {int(torch.isnan(cur).sum())}")
            # finer: within the first differing tile, which (co % 64, t % 128)?
            c0, t0 = int(cs[0]) // 64, int(ts[0]) // 128
            m = d[idx[0, 0], c0 * 64:(c0 + 1) * 64, t0 * 128:(t0 + 1) * 128]
            print("   tile rows with differences:", sorted(set(m.nonzero()[:, 0].tolist()))[:70])
            print("   tile cols with differences:", sorted(set(m.nonzero()[:, 1].tolist()))[:140])
        else:
            print(f"rep {rep}: equal")
finally:
    child.kill()
