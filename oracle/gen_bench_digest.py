"""Oracle digest of the BENCHMARKED workload (test infrastructure; imported by nothing in the product path).

bench.py draws its router noise on the device (counter-based stream keyed by seed / global clip / evaluation / block / gate);
versband_amd.prng.device_router_exponentials restates that stream on the host, so the CPU oracle can replay clip 0 of
bench.py's first timed pass exactly: 50 CFG Euler steps at T = 752, L = 80, scale 3, then the VAE decode.  The result is
committed as tests/golden/bench_clip0.npz (the oracle's latent in full + a digest of its mel) and bench.py / the GPU tests
compare the HIP path's bf16 production output with it at north_star's tolerances.

    python oracle/gen_bench_digest.py            # ~1-2 min of CPU: configs[1] (E = 4) -> tests/golden/bench_clip0.npz
    python oracle/gen_bench_digest.py --experts 8   # configs[2] (Band-MoE stress, E = 8) -> tests/golden/bench_clip0_e8.npz
    python oracle/gen_bench_digest.py --long        # configs[4] (120 s long-form, batch 4: clip 0 = 4 windows of 1500) -> bench_clip0_long.npz, ~5 min
    python oracle/gen_bench_digest.py --checks      # everything bench.py's default run verifies beyond clip 0 of pass 0 (round 3):
                                                    # clips 0 and 4 (the first clip of each of the two sub-batches / streams) x passes 0 and 1
                                                    # (seed 1234 and 1235: pass 1 is the first hipGraph REPLAY of the timed region), plus the
                                                    # oracle's states x_k and routing indices of clip 0 / pass 0 at six steps for the
                                                    # teacher-forced routing-flip count -> tests/golden/bench_c2_checks.npz, ~10 min of CPU
    python oracle/gen_bench_digest.py --checks-add 7   # round 4: add global clip 7 (last row of the second sub-batch) x passes 0 and 1 to it
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu  # noqa: E402
from tests.helpers import clip_batch  # noqa: E402
from versband_amd import prng, synth  # noqa: E402

SEED, T, L, E, DEPTH, STEPS, SCALE = 1234, 752, 80, 4, 4, 50, 3.0


def digest(a, prefix, n_samples=256):
    """fingerprint in the key layout tests/helpers.check_digest reads (same definition as oracle/gen_golden.py:digest)"""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = (np.arange(n_samples, dtype=np.int64) * 2654435761 + 12345) % a.size
    d = {"shape_numel": np.array([a.size], dtype=np.int64), "sum": np.array([a.sum()]), "l2": np.array([np.sqrt((a * a).sum())]),
         "maxabs": np.array([np.abs(a).max()]), "idx": idx, "val": a[idx]}
    return {prefix + k: v for k, v in d.items()}


def device_noise(seed, clip, nfe, branch, E=E, T=T):
    out = []
    for blk in range(DEPTH):
        out.append(tuple(torch.from_numpy(prng.device_router_exponentials(seed, clip, nfe, branch, blk, gate, T, w))
                         for gate, w in ((0, 2), (1, E), (2, E))))
    return out


def main_long():
    """bench.py --workload c5: 4 clips of 120 s (T = 4500) per GPU; versband_amd.longform.sample_long turns the windows into batch rows
    (row = window * B + clip, noise key = clip_base * n_windows + row) and cross-fades them.  Clip 0 = rows 0, 4, 8, 12."""
    # (the window plan and the cross-fade are restated HERE: the oracle must not lean on the product package for the thing it checks)
    def plan_windows(T, window, overlap):
        """[(start, length)]: windows of `window` tokens hopping by window - overlap, the last one aligned to the end of the clip"""
        if T <= window:
            return [(0, T)]
        out, s0 = [], 0
        while s0 + window < T:
            out.append((s0, window))
            s0 += window - overlap
        out.append((T - window, window))
        return out

    def crossfade_windows(parts, plan, T):
        """linear ramps over every overlap, weights normalised to one"""
        acc = torch.zeros(parts[0].shape[0], parts[0].shape[1], T, dtype=parts[0].dtype)
        wsum = torch.zeros(T, dtype=parts[0].dtype)
        for i, ((s0, n), part) in enumerate(zip(plan, parts)):
            w = torch.ones(n, dtype=part.dtype)
            if i > 0:
                ov = plan[i - 1][0] + plan[i - 1][1] - s0
                if ov > 0:
                    w[:ov] = torch.linspace(0, 1, ov + 2, dtype=part.dtype)[1:-1]
            if i + 1 < len(plan):
                ov = s0 + n - plan[i + 1][0]
                if ov > 0:
                    w[n - ov:] = torch.minimum(w[n - ov:], torch.linspace(1, 0, ov + 2, dtype=part.dtype)[1:-1])
            acc[:, :, s0:s0 + n] += part * w
            wsum[s0:s0 + n] += w
        return acc / wsum
    B, TL, WIN, OV = 4, 4500, 1500, 128
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    dcfg, vcfg = synth.DiTConfig(), synth.VAEConfig()
    sd = synth.make_state_dict(synth.dit_shapes(dcfg), SEED)
    sdv = synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1)
    inp = clip_batch(1, TL, L, clip0=0, seed=SEED)
    plan = plan_windows(TL, WIN, OV)
    midi, beats = inp["midi"].reshape(1, -1), inp["beats"].reshape(1, -1)
    parts = []
    for w, (s0, n) in enumerate(plan):
        key = w * B + 0
        cc = ref_cpu.dit_precompute(sd, inp["t5_cond"], midi[:, 2 * s0:2 * (s0 + n)], beats[:, 2 * s0:2 * (s0 + n)], n)
        cu = ref_cpu.dit_precompute(sd, inp["t5_uncond"], midi[:, 2 * s0:2 * (s0 + n)], beats[:, 2 * s0:2 * (s0 + n)], n)
        parts.append(ref_cpu.sample_cfg(sd, inp["x_latent"][:, :, s0:s0 + n], cc, cu, SCALE, STEPS + 1,
                                        lambda k, br, key=key, n=n: device_noise(SEED, key, k, br, 4, n)))
        print("window", w, (s0, n), "done", flush=True)
    z = crossfade_windows(parts, plan, TL)
    mel = ref_cpu.vae_decode(sdv, z)
    out = {"meta": np.array([SEED, TL, L, 4, STEPS, B, WIN, OV], dtype=np.int64), "scale": np.float32(SCALE), "z": z.numpy()}
    out.update(digest(mel, "mel_"))
    path = os.path.join(ROOT, "tests", "golden", "bench_clip0_long.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


TF_STEPS = (0, 10, 20, 30, 40, 49)     # Euler steps whose oracle state + routing indices are kept for the teacher-forced flip count


def main_checks(add_clips=None):
    """clip c of pass p of `python bench.py` = global clip index c, sampler seed SEED + p (bench.py:one_pass), router noise keyed by
    (seed, global clip, evaluation index = Euler step, branch, block, gate).
    add_clips (--checks-add 7): keep the committed entries and add these clips (round 4: global clip 7 = the last row of the second
    sub-batch, i.e. the partial-tile edge of every token-row launch)."""
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    dcfg, vcfg = synth.DiTConfig(num_experts=E), synth.VAEConfig()
    sd = synth.make_state_dict(synth.dit_shapes(dcfg), SEED)
    sdv = synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1)
    path = os.path.join(ROOT, "tests", "golden", "bench_c2_checks.npz")
    if add_clips:
        out = dict(np.load(path))
        out["clips"] = np.array(sorted(set(int(c) for c in out["clips"]) | set(add_clips)), dtype=np.int64)
        todo = tuple(add_clips)
    else:
        out = {"meta": np.array([SEED, T, L, E, STEPS], dtype=np.int64), "scale": np.float32(SCALE), "clips": np.array([0, 4], dtype=np.int64),
               "passes": np.array([0, 1], dtype=np.int64), "tf_steps": np.array(TF_STEPS, dtype=np.int64)}
        todo = (0, 4)
    for clip in todo:
        inp = clip_batch(1, T, L, clip0=clip, seed=SEED)
        cc = ref_cpu.dit_precompute(sd, inp["t5_cond"], inp["midi"], inp["beats"], T)
        cu = ref_cpu.dit_precompute(sd, inp["t5_uncond"], inp["midi"], inp["beats"], T)
        for ps in (0, 1):
            seed = SEED + ps
            tf = clip == 0 and ps == 0
            t_span, idx = ref_cpu.t_index_table(STEPS + 1)
            x = inp["x_latent"].clone()
            t = t_span[0]
            for k in range(STEPS):                      # ref_cpu.sample_cfg unrolled (same bookkeeping) so states and routes can be kept
                dt = t_span[k + 1] - t
                ti = torch.full((1,), int(idx[k]), dtype=torch.long)
                nz = [device_noise(seed, clip, k, br, E) for br in (0, 1)]
                if tf and k in TF_STEPS:
                    out[f"tf_x_{k}"] = x.numpy().copy()
                    e_c, aux_c = ref_cpu.dit_forward(sd, x, ti, cc, nz[0], return_aux=True)
                    e_u, aux_u = ref_cpu.dit_forward(sd, x, ti, cu, nz[1], return_aux=True)
                    r = np.zeros((DEPTH, 2, 2 * T), dtype=np.int8)     # [block][gate: caption, acoustic][cond rows then uncond rows]
                    for br, aux in ((0, aux_c), (1, aux_u)):
                        for i in range(DEPTH):
                            r[i, 0, br * T:(br + 1) * T] = aux[f"ic{i}"].reshape(-1).numpy()
                            r[i, 1, br * T:(br + 1) * T] = aux[f"ia{i}"].reshape(-1).numpy()
                    out[f"tf_routes_{k}"] = r
                    out[f"tf_tidx_{k}"] = np.array([int(idx[k])], dtype=np.int64)
                else:
                    e_c = ref_cpu.dit_forward(sd, x, ti, cc, nz[0])
                    e_u = ref_cpu.dit_forward(sd, x, ti, cu, nz[1])
                x = x + dt * (e_u + SCALE * (e_c - e_u))
                t = t + dt
            mel = ref_cpu.vae_decode(sdv, x)
            out[f"z_c{clip}_p{ps}"] = x.numpy()
            out.update(digest(mel, f"mel_c{clip}_p{ps}_"))
            print(f"clip {clip} pass {ps} done", flush=True)
    np.savez_compressed(path, **out)
    print("wrote", path)


def main():
    if "--long" in sys.argv:
        return main_long()
    if "--checks-add" in sys.argv:
        return main_checks([int(v) for v in sys.argv[sys.argv.index("--checks-add") + 1].split(",")])
    if "--checks" in sys.argv:
        return main_checks()
    E = int(sys.argv[sys.argv.index("--experts") + 1]) if "--experts" in sys.argv else 4
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    dcfg, vcfg = synth.DiTConfig(num_experts=E), synth.VAEConfig()
    sd = synth.make_state_dict(synth.dit_shapes(dcfg), SEED)
    sdv = synth.make_state_dict(synth.vae_decoder_shapes(vcfg), SEED + 1)
    inp = clip_batch(1, T, L, clip0=0, seed=SEED)
    cc = ref_cpu.dit_precompute(sd, inp["t5_cond"], inp["midi"], inp["beats"], T)
    cu = ref_cpu.dit_precompute(sd, inp["t5_uncond"], inp["midi"], inp["beats"], T)
    z = ref_cpu.sample_cfg(sd, inp["x_latent"], cc, cu, SCALE, STEPS + 1, lambda k, br: device_noise(SEED, 0, k, br, E))
    mel = ref_cpu.vae_decode(sdv, z)
    out = {"meta": np.array([SEED, T, L, E, STEPS], dtype=np.int64), "scale": np.float32(SCALE), "z": z.numpy()}
    out.update(digest(mel, "mel_"))
    path = os.path.join(ROOT, "tests", "golden", "bench_clip0.npz" if E == 4 else f"bench_clip0_e{E}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
