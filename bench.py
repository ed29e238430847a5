#!/usr/bin/env python3
"""Headline benchmark: mel-seconds generated per second (BASELINE.json metric).

A "step" = one pass of the whole hot path over one batch of synthetic clips that are already
resident in HBM: conditioning precompute -> 50 Euler flow steps with CFG (2 network evaluations
per step, batched) -> VAE decode -> full HiFi-GAN decode.  Workload = BASELINE.json configs[1]:
batch 8 x 20 s clips, bf16 DiT + fp32 VAE/vocoder, random-init checkpoints of the configured
architecture (synthetic, no network).  N>1: one process per GPU, clips sharded by rank (weak
scaling), weights broadcast once from rank 0 over RCCL, no data-path collective.

    python bench.py --gpus 1 --steps 2 --warmup 1
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

CLIP_SECONDS = 20.0
T_LAT, L_CTX = 752, 80
SEED = 1234
PEAK = {0: ("mfma", 2500.0, "bf16 MFMA GEMM (DiT projections + experts)"),
        1: ("mfma", 2500.0, "bf16 flash attention"),
        2: ("mfma", 2500.0 / 3.0, "split-bf16 (bf16x3) MFMA implicit-GEMM conv1d (VAE + HiFi-GAN); peak = bf16 dense / 3 passes")}


_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """threads the CPU baseline may use: affinity mask, capped by the cgroup CPU quota (a container that
    reports 256 CPUs but is throttled to a few would otherwise spin 256 OpenMP threads on them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, 64))


CLASS_KERNELS = {0: ("gemm_bf16",), 1: ("attn_kernel",), 2: ("conv1d_", "respair_")}


def pmc_traffic(cls):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 PMC passes (profiles/r01_pmc_summary.json:
    separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command; counter unit = KB; FETCH_SIZE doubled for
    gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        tot, n = 0.0, 0
        for name, v in d.get("FETCH_SIZE", {}).items():
            if any(k in name for k in CLASS_KERNELS[cls]):
                tot += 2.0 * v["sum_kb"] * 1024.0
                n += v["launches"]
        for name, v in d.get("WRITE_SIZE", {}).items():
            if any(k in name for k in CLASS_KERNELS[cls]):
                tot += v["sum_kb"] * 1024.0
        return (tot / n) if n else None
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--flow-steps", type=int, default=50)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "split"])
    ap.add_argument("--scale", type=float, default=3.0)
    ap.add_argument("--streams", type=int, default=2, help="independent sub-batches per GPU, one HIP stream + host thread each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-flow-steps", type=int, default=4)
    ap.add_argument("--cpu-timeout", type=float, default=240.0)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--save-out", default=None, help="directory: every rank writes the waveforms of its last pass (clip-indexed .npy)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` on its own: start one process per GPU, like the reference does with
    mp.spawn(gen_song, nprocs=num_gpus) (scripts/test_final.py:467-477), by re-running this file under
    torch.distributed.run (rendezvous on 127.0.0.1).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    log(f"spawning {args.gpus} ranks: {' '.join(cmd[1:8])} ...")
    return subprocess.call(cmd, env=env)


def broadcast_state(sds, rank, world, device):
    """rank 0's checkpoints -> every rank, one flat RCCL broadcast (the only collective of the path)."""
    import torch.distributed as dist
    flat_keys = [(i, k) for i, sd in enumerate(sds) for k in sd]
    sizes = [sds[i][k].numel() for i, k in flat_keys]
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if rank == 0:
        off = 0
        for (i, k), n in zip(flat_keys, sizes):
            buf[off:off + n].copy_(sds[i][k].reshape(-1))
            off += n
    dist.broadcast(buf, src=0)
    off = 0
    out = [dict() for _ in sds]
    for (i, k), n in zip(flat_keys, sizes):
        out[i][k] = buf[off:off + n].view(sds[i][k].shape)
        off += n
    return out


def cpu_baseline(sd_d, sd_v, sd_h, hp, flow_steps_total, cpu_steps, scale):
    """The CPU oracle (a port of the reference's algorithm, routed + hoisted variant) on the host cores:
    B=1, one 20 s clip; `cpu_steps` Euler steps are timed and extrapolated linearly to the full count,
    VAE decode and HiFi-GAN are timed in full."""
    from oracle import ref_cpu
    from tests.helpers import clip_batch, exp_noise
    from versband_amd import model as vm
    cores = usable_cores()
    torch.set_num_threads(cores)
    inp = clip_batch(1, T_LAT, L_CTX, clip0=0, seed=SEED)
    t0 = time.perf_counter()
    cc = ref_cpu.dit_precompute(sd_d, inp["t5_cond"], inp["midi"], inp["beats"], T_LAT)
    cu = ref_cpu.dit_precompute(sd_d, inp["t5_uncond"], inp["midi"], inp["beats"], T_LAT)
    t_pre = time.perf_counter() - t0
    noise = {(k, br): exp_noise(1, T_LAT, 4, 2 * k + br, 4, seed=SEED) for k in range(cpu_steps) for br in (0, 1)}
    t0 = time.perf_counter()
    z = ref_cpu.sample_cfg(sd_d, inp["x_latent"], cc, cu, scale, flow_steps_total + 1,
                           lambda k, br: noise[(k, br)]) if cpu_steps >= flow_steps_total else None
    if z is None:      # bounded sample: run cpu_steps Euler steps of the same schedule
        t_span, idx = ref_cpu.t_index_table(flow_steps_total + 1)
        x = inp["x_latent"].clone()
        for k in range(cpu_steps):
            ti = torch.full((1,), idx[k], dtype=torch.long)
            e_c = ref_cpu.dit_forward(sd_d, x, ti, cc, noise[(k, 0)])
            e_u = ref_cpu.dit_forward(sd_d, x, ti, cu, noise[(k, 1)])
            x = x + (t_span[k + 1] - t_span[k]) * (e_u + scale * (e_c - e_u))
        z = x
    t_flow = (time.perf_counter() - t0) * (flow_steps_total / float(min(cpu_steps, flow_steps_total)))
    t0 = time.perf_counter()
    mel = ref_cpu.vae_decode(sd_v, z)
    t_vae = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref_cpu.hifigan_forward(sd_h, hp, mel)
    t_voc = time.perf_counter() - t0
    total = t_pre + t_flow + t_vae + t_voc
    # the reference as written (second variant): all 3E expert FFNs on every token and the whole conditioning stem
    # recomputed inside each of the two sequential network evaluations of a step - one step timed, scaled to the full count
    t0 = time.perf_counter()
    ti = torch.full((1,), 0, dtype=torch.long)
    for br, t5 in ((0, inp["t5_cond"]), (1, inp["t5_uncond"])):
        cnd = ref_cpu.dit_precompute(sd_d, t5, inp["midi"], inp["beats"], T_LAT)
        ref_cpu.dit_forward(sd_d, inp["x_latent"], ti, cnd, noise[(0, br)], dense=True)
    t_dense = (time.perf_counter() - t0) * flow_steps_total
    dense = {"value": CLIP_SECONDS / (t_dense + t_vae + t_voc), "unit": "mel-s/s",
             "what": f"reference-faithful evaluation order: dense experts + conditioning recomputed per evaluation, 1 of {flow_steps_total} "
                     f"steps timed and scaled to {t_dense:.1f}s, same VAE / HiFi-GAN times"}
    return {"value": CLIP_SECONDS / total, "unit": "mel-s/s", "cores": cores, "kind": "port", "reference_faithful": dense,
            "sample": f"B=1 x 20 s clip on {cores} threads: cond precompute {t_pre:.2f}s + {min(cpu_steps, flow_steps_total)} of "
                      f"{flow_steps_total} CFG Euler steps timed and scaled to {t_flow:.1f}s + full VAE decode {t_vae:.2f}s + "
                      f"full HiFi-GAN {t_voc:.2f}s (torch fp32 oracle, routed experts, conditioning hoisted)"}


def cpu_baseline_subprocess(args):
    """run the CPU leg in its own process with a hard time bound so the bench can never hang on it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--flow-steps", str(args.flow_steps),
           "--cpu-flow-steps", str(args.cpu_flow_steps), "--scale", str(args.scale)]
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "mel-s/s", "cores": usable_cores(), "kind": "port", "sample": "failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "mel-s/s", "cores": usable_cores(), "kind": "port",
                "sample": f"timed out after {args.cpu_timeout:.0f}s"}


def cpu_worker(args):
    from versband_amd import synth
    dcfg, vcfg, hcfg = synth.DiTConfig(), synth.VAEConfig(), synth.HifiGanConfig()
    sds = [synth.make_state_dict(s, SEED + i) for i, s in
           enumerate([synth.dit_shapes(dcfg), synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)])]
    print(json.dumps(cpu_baseline(sds[0], sds[1], sds[2], hcfg.as_hparams(), args.flow_steps, args.cpu_flow_steps, args.scale)))


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_worker(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} rank(s): one process per GPU is the contract"
    one_device = bool(os.environ.get("VB_BENCH_ONE_DEVICE"))
    if not one_device:
        assert torch.cuda.device_count() >= world, f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()}"
    if os.environ.get("VB_BENCH_ONE_DEVICE"):          # functional test of the N > 1 code path on a 1-GPU box (gloo, all ranks on cuda:0)
        local = 0
    assert torch.cuda.is_available(), "bench.py measures the HIP path: no GPU visible"
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("VB_BENCH_ONE_DEVICE"):
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    from tests.helpers import clip_batch
    from versband_amd import _lib as L
    from versband_amd import model as vm
    from versband_amd import synth
    from versband_amd.engine import Context, DiTEngine, build_hifigan, build_vae_decoder

    dcfg, vcfg, hcfg = synth.DiTConfig(), synth.VAEConfig(), synth.HifiGanConfig()
    shapes = [synth.dit_shapes(dcfg), synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)]
    if rank == 0:
        sds = [synth.make_state_dict(s, SEED + i) for i, s in enumerate(shapes)]
    else:
        sds = [{k: torch.empty(shp) for k, (shp, _) in s.items()} for s in shapes]
    bcast_ms, bcast_bytes = None, 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        sds = broadcast_state(sds, rank, world, device)
        torch.cuda.synchronize()
        bcast_ms = 1e3 * (time.perf_counter() - t0)
        bcast_bytes = 4 * sum(v.numel() for sd in sds for v in sd.values())
        seen = torch.ones(1, device=device)
        dist.all_reduce(seen)                      # every rank reports in over the data-path backend (RCCL unless one-device test)
        assert int(seen.item()) == world
    log("weights ready; packing")
    ctx = Context(device)
    S = max(1, args.streams)
    B = args.batch
    assert B % S == 0, "--batch must be divisible by --streams"
    Bs = B // S
    idx, dts = vm.euler_tables(args.flow_steps + 1)
    # S independent sub-batches, each with its own engine handles, on its own HIP stream driven by its own host thread:
    # kernels of different sub-batches overlap on the GPU and fill each other's tile-quantisation tails.
    workers = []
    for si in range(S):
        eng = DiTEngine(ctx, dcfg, sds[0], precision=args.precision)
        vae = build_vae_decoder(ctx, sds[1])
        voc = build_hifigan(ctx, sds[2], hcfg.as_hparams())
        inp = clip_batch(Bs, T_LAT, L_CTX, clip0=rank * B + si * Bs, seed=SEED)
        workers.append(dict(eng=eng, vae=vae, voc=voc, x0=inp["x_latent"].to(device),
                            t5=torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(device), midi=inp["midi"].to(device),
                            beats=inp["beats"].to(device), stream=torch.cuda.Stream(device=device), clip_base=rank * B + si * Bs,
                            wav=None))

    def run_worker(w, ks):
        torch.cuda.set_device(device)           # a new host thread starts on device 0: bind it to this rank's GPU
        with torch.cuda.stream(w["stream"]):
            for k in ks:
                cond = w["eng"].precompute_cond(w["t5"], w["midi"], w["beats"], T_LAT)
                z = w["eng"].sample_cfg(w["x0"], cond, idx, dts, args.scale, seed=SEED + k, clip_base=w["clip_base"])
                w["wav"] = w["voc"].run(w["vae"].run(z))

    def run_passes(ks):
        import threading
        if S == 1:
            run_worker(workers[0], ks)
        else:
            ths = [threading.Thread(target=run_worker, args=(w, ks)) for w in workers]
            for t in ths:
                t.start()
            for t in ths:
                t.join()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    lib = L.load()
    torch.cuda.synchronize()
    log(f"engines built ({S} stream(s) x {Bs} clips); warmup")

    def read_prof(cls):
        ms, fl, n, nt = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
        L.check(lib.vb_prof_read(cls, C.byref(ms), C.byref(fl), C.byref(n), C.byref(nt)), "vb_prof_read")
        return ms.value, fl.value, n.value, nt.value

    # warmup; the last warmup pass times (a sample of) every kernel class to find the dominant one
    EVERY = 5            # coprime with the 4 GEMMs / block so the sample walks over every kernel of the class
    dominant, breakdown = 0, {}
    for wi in range(max(args.warmup, 1)):
        last = wi == max(args.warmup, 1) - 1
        if last:
            L.check(lib.vb_prof_enable(7 | (EVERY << 8)), "prof")
        run_passes([-1 - wi])
        torch.cuda.synchronize()
        if last:
            for cls in (0, 1, 2):
                ms, fl, n, nt = read_prof(cls)
                breakdown[cls] = ms * (n / nt) if nt else 0.0      # sampled launches extrapolated to the class
            dominant = max(breakdown, key=breakdown.get)
    assert all(torch.isfinite(w["wav"]).all() for w in workers)
    # the dominant class alone on the GPU (one stream, the whole batch of B clips): the kernel-quality figure that the
    # concurrent timed region dilutes (two sub-batches share the CUs, so each launch is slower but two are in flight)
    isolated = None
    if S > 1:
        inp = clip_batch(B, T_LAT, L_CTX, clip0=rank * B, seed=SEED)
        w = dict(eng=DiTEngine(ctx, dcfg, sds[0], precision=args.precision), vae=build_vae_decoder(ctx, sds[1]),
                 voc=build_hifigan(ctx, sds[2], hcfg.as_hparams()), x0=inp["x_latent"].to(device),
                 t5=torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(device), midi=inp["midi"].to(device),
                 beats=inp["beats"].to(device), stream=torch.cuda.Stream(device=device), clip_base=rank * B, wav=None)
        run_worker(w, [-100])
        torch.cuda.synchronize()
        L.check(lib.vb_prof_enable((1 << dominant) | (EVERY << 8)), "prof")
        run_worker(w, [-101])
        torch.cuda.synchronize()
        ms, fl, n, nt = read_prof(dominant)
        if ms > 0 and nt > 0:
            isolated = {"achieved": fl / (ms * 1e-3) / 1e12, "avg_launch_us": 1e3 * ms / nt, "timed_launches": nt,
                        "what": f"same class, one stream, one batch of {B} clips (nothing else on the GPU)"}
        del w
        torch.cuda.empty_cache()
    L.check(lib.vb_prof_enable((1 << dominant) | (EVERY << 8)), "prof")
    log(f"warmup done; class ms/pass = {breakdown}; timing {args.steps} step(s)")

    barrier()
    t0 = time.perf_counter()
    run_passes(list(range(args.steps)))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    log(f"timed region: {elapsed:.3f}s")
    ms, fl, n, nt = read_prof(dominant)
    L.check(lib.vb_prof_enable(0), "prof")

    if rank == 0:
        bound, peak, kname = PEAK[dominant]
        achieved = (fl / (ms * 1e-3) / 1e12) if (ms > 0 and nt > 0) else 0.0      # flops and time of the timed launches
        total_mel_s = world * B * CLIP_SECONDS * args.steps
        out = {
            "metric": "mel-seconds generated/sec (20 s clip, %d flow steps)" % args.flow_steps,
            "value": total_mel_s / elapsed,
            "unit": "mel-s/s",
            "n_gpus": world,
            "ranks": {"world": world, "backend": ("gloo (VB_BENCH_ONE_DEVICE functional test)" if os.environ.get("VB_BENCH_ONE_DEVICE") else
                                                 "nccl (RCCL)") if world > 1 else None,
                      "weight_broadcast_ms": bcast_ms, "weight_broadcast_bytes": bcast_bytes, "collectives_in_timed_region": 0},
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("bf16 DiT (fp32 accumulate)" if args.precision == "bf16" else "bf16x3 split DiT") +
                     " + fp32-I/O VAE/vocoder on split-bf16 (bf16x3) MFMA, <=3e-5 of exact fp32",
            "data": "synthetic (seeded PRNG clips, random-init checkpoints of the configured architecture)",
            "config": {"workload": f"{B} x 20 s clips per GPU (T_lat=752, T_mel=1504, 24 kHz), {args.flow_steps} Euler steps x 2 NFE (CFG "
                                   f"scale {args.scale}), Band-MoE E=4, VAE decode + HiFi-GAN V1-like (8*5*4*2), configs/vocal2music.yaml",
                       "clips_per_gpu": B, "flow_steps": args.flow_steps, "precision": args.precision,
                       "streams_per_gpu": S, "parallelism": f"batch-shard x{world} ({S} concurrent sub-batches of {Bs} clips per GPU)"},
            "roofline": {"bound": bound, "kernel": kname, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": pmc_traffic(dominant),
                         "traffic_note": "HBM bytes per launch averaged over the class, from the committed rocprofv3 --pmc FETCH_SIZE / "
                                         "WRITE_SIZE passes of this command (profiles/r01_pmc_summary.json); algorithmic bytes in DESIGN.md",
                         "avg_launch_us": (1e3 * ms / nt) if nt else None, "launches_per_step": n / max(args.steps, 1),
                         "timed_launches": nt, "note": "every 5th launch of the class is bracketed by HIP events on its own stream; with "
                         "streams_per_gpu > 1 a launch shares the GPU with the other sub-batch's kernels, so its duration includes that overlap",
                         "isolated": (dict(isolated, frac=isolated["achieved"] / peak) if isolated else None),
                         "class_ms_per_step_warmup": {PEAK[c][2]: round(v, 3) for c, v in breakdown.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (subprocess, bounded)")
            out["cpu_baseline"] = cpu_baseline_subprocess(args)
        print(json.dumps(out))
    if args.save_out:
        import numpy as np
        os.makedirs(args.save_out, exist_ok=True)
        for w in workers:
            wav = w["wav"].detach().cpu().numpy()
            for i in range(wav.shape[0]):
                np.save(os.path.join(args.save_out, f"clip{w['clip_base'] + i:04d}.npy"), wav[i])
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
