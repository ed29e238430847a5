#!/usr/bin/env python3
"""Headline benchmark: mel-seconds generated per second (BASELINE.json metric).

A "step" = one pass of the whole hot path over one batch of synthetic clips that are already
resident in HBM: conditioning precompute -> Euler flow steps with CFG (2 network evaluations
per step, batched) -> VAE decode -> full HiFi-GAN decode.

  --workload c2 (default)  BASELINE configs[1]: batch 8 x 20 s clips, 50 flow steps, bf16 DiT + fp32 VAE/vocoder (`value`: fp32 products on the f32 MFMA,
                           F(2,3) minimal filtering where it applies); the same passes with the direct fp32 kernels (`fp32_direct`) and the
                           bf16x3 VAE/vocoder (`split`) are timed in their own regions
  --workload c3            BASELINE configs[2]: Band-MoE stress, num_experts = 8, batch 32 (48 128 token rows per CFG branch)
  --workload c5            BASELINE configs[4]: long-form, batch 4 x 120 s (T = 4500 latent frames in windows of 1500 tokens,
                           cross-faded) + VAE decode of the whole latent + halo'd chunked vocoding

N > 1: one process per GPU (`python bench.py --gpus N` spawns them itself; under torch.distributed.run it joins the ranks it
is given), clips sharded by rank (weak scaling), weights broadcast once from rank 0 over RCCL, no data-path collective.

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

L_CTX = 80
SEED = 1234
HBM_PEAK_TBS = 8.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s float4-copy rate measured)
MFMA_BF16_TF = 2500.0       # dense bf16 MFMA peak
MFMA_F32_TF = 157.3         # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32), the roof of --vocoder-precision fp32
# kernel classes of the library's HIP-event profiler (csrc/kernels.h): name, binding roof, peak TFLOP/s
CLASSES = {
    0: ("bf16 MFMA GEMM (DiT projections, routed + band experts)", "mfma", MFMA_BF16_TF),
    1: ("bf16 flash attention (self + T5 cross)", "mfma", MFMA_BF16_TF),
    2: ("split-bf16 (bf16x3) MFMA implicit-GEMM conv1d (VAE + HiFi-GAN)", "mfma", MFMA_BF16_TF / 3.0),
    3: ("fused HiFi-GAN ResBlock pair (bf16x3 MFMA, intermediate in LDS)", "mfma|hbm", MFMA_BF16_TF / 3.0),
}
CLASSES_FP32 = {
    2: ("fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit-GEMM conv1d (VAE + HiFi-GAN)", "mfma", MFMA_F32_TF),
    3: ("fused HiFi-GAN ResBlock pair (f32 MFMA, intermediate in LDS)", "mfma|hbm", MFMA_F32_TF),
}
CLASS_KERNELS = {0: ("gemm_bf16", "band_ffn", "moe_ffn", "moe_w2"), 1: ("attn_kernel",), 2: ("conv1d_",), 3: ("respair_",)}


def classes_for(vocoder_precision):
    c = dict(CLASSES)
    if vocoder_precision in ("fp32", "fp32mf"):
        c.update(CLASSES_FP32)
    return c

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


LINE_MAX = 4096             # the driver reads the LAST stdout line as JSON; round 5's 21 KB line came back unparsed


def _r(x, nd=6):
    """round floats for the one-line summary (the side file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    return x


def compact_line(out: dict, detail_path: str | None) -> str:
    """The ONE stdout JSON line of the bench contract, <= LINE_MAX bytes: the contract keys, `roofline`, `cpu_baseline`, a
    parity verdict and the second region's headline.  Everything tabular (per-class / per-group / per-kernel tables, the
    verified clip list, device telemetry, prose) goes to `detail_path` (bench_detail.json), which the line names."""
    rf = out.get("roofline") or {}
    pr = rf.get("path_roofline") or {}
    pc = out.get("parity_check")
    cfg = out.get("config") or {}
    line = {k: _r(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                         "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "baseline_config", "clips_per_gpu", "clip_seconds", "flow_steps", "precision", "experts",
                                              "vocoder_precision", "streams_per_gpu", "parallelism", "sampler_loop") if k in cfg}
    if pc is not None:
        worst = None
        cp = (pc.get("verified") or {}).get("clips_x_passes") or [pc]
        for key in ("latent_rel_l2", "mel_l1_sampled", "mel_l2_rel"):
            v = [c[key] for c in cp if isinstance(c.get(key), (int, float))]
            if v:
                worst = worst or {}
                worst[key] = _r(max(v), 3)
        rfp = pc.get("routing_flips") or {}
        line["parity_check"] = {"ok": pc.get("ok"), "tol": pc.get("tol"), "worst": worst, "clips_x_passes_checked": len(cp),
                                "routing_flips_per_block": rfp.get("per_block"), "routing_decisions_per_block": rfp.get("decisions_per_block")}
    line["roofline"] = {"bound": rf.get("bound"), "kernel": rf.get("kernel"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"),
                        "unit": rf.get("unit"), "frac": _r(rf.get("frac"), 4), "traffic": _r(rf.get("traffic")),
                        "algorithmic_bytes_per_launch": _r(rf.get("algorithmic_bytes_per_launch")), "avg_launch_us": _r(rf.get("avg_launch_us"), 5),
                        "path_frac": _r(pr.get("frac"), 4), "path_ideal_ms": _r(pr.get("ideal_ms_per_pass"), 5),
                        "path_frac_direct_equivalent": _r(pr.get("direct_equivalent_frac"), 4),
                        "groups": [{"group": g["group"].split(" (")[0][:48], "ms_per_pass": _r(g.get("ms_per_pass"), 4), "frac": _r(g.get("frac_of_mfma_peak"), 3)}
                                   for g in (rf.get("groups") or [])]}
    cb = out.get("cpu_baseline")
    if cb is not None:
        rfa = cb.get("reference_faithful") or {}
        line["cpu_baseline"] = {"value": _r(cb.get("value"), 4), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": (cb.get("sample") or "")[:300], "reference_faithful": _r(rfa.get("value"), 4)}
        if cb.get("error"):
            line["cpu_baseline"]["error"] = str(cb["error"])[:200]
    for key in ("fp32_direct", "split"):
        sp = out.get(key)
        if sp is not None:
            line[key] = {"value": _r(sp.get("value")), "ms_per_step": _r(sp.get("ms_per_step")), "vocoder_precision": sp.get("vocoder_precision"),
                         "parity_ok": (sp.get("parity_check") or {}).get("ok")}
    deq = rf.get("direct_equivalent")
    if deq:
        line["roofline"]["direct_equivalent"] = {"achieved": _r(deq.get("achieved")), "frac": _r(deq.get("frac"), 4)}
    rk = out.get("ranks") or {}
    if (rk.get("world") or 1) > 1:
        line["ranks"] = {"backend": rk.get("backend"), "weight_broadcast_ms": _r(rk.get("weight_broadcast_ms"), 4),
                         "weight_broadcast_gbps": _r(rk.get("weight_broadcast_gbps"), 4), "per_rank_ms": [_r(x, 5) for x in (rk.get("per_rank_ms") or [])][:8],
                         "collectives_in_timed_region": rk.get("collectives_in_timed_region")}
    line["detail"] = detail_path
    s = json.dumps(line, separators=(",", ":"))
    # belt and braces: shed optional blocks before ever exceeding the limit (never the contract keys, roofline or cpu_baseline)
    for drop in ("ranks", "split", "fp32_direct", "data"):
        if len(s) <= LINE_MAX:
            break
        line.pop(drop, None)
        s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_MAX:
        line["config"] = {"workload": str(cfg.get("workload"))[:200]}
        line["roofline"].pop("groups", None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= LINE_MAX and "\n" not in s, len(s)
    return s


def _class_patterns(cls):
    """kernel-name patterns of a class id or of a group (tuple) of class ids"""
    ids = cls if isinstance(cls, (tuple, list)) else (cls,)
    return tuple(p for c in ids for p in CLASS_KERNELS[c])


def pmc_traffic(cls):
    """HBM bytes per launch of a kernel class from the newest committed rocprofv3 PMC summary (profiles/r*_pmc_summary.json:
    separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this same command; counter unit = KB; FETCH_SIZE doubled for gfx950 as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside the timed process, so this figure
    comes from a FILE and the JSON line says so.  (None, None) when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    path = files[-1]
    try:
        d = json.load(open(path))
        tot, n = 0.0, 0
        for name, v in d.get("FETCH_SIZE", {}).items():
            if any(k in name for k in _class_patterns(cls)):
                tot += 2.0 * v["sum_kb"] * 1024.0
                n += v["launches"]
        for name, v in d.get("WRITE_SIZE", {}).items():
            if any(k in name for k in _class_patterns(cls)):
                tot += v["sum_kb"] * 1024.0
        return ((tot / n) if n else None), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def pmc_traffic_live(cls, timeout, vocoder_precision="fp32", launches_per_pass=None):
    """HBM bytes per launch of a kernel class MEASURED IN THIS RUN: two child runs of this same command (one stream, one pass, no
    baseline legs) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes: the two counters do
    not fit the TCC's 4 slots together), counter unit = KB, FETCH_SIZE doubled for gfx950 as /opt/skills/guides/MI355X_MICROARCH.md
    prescribes.  Returns (bytes per launch | None, how, per-kernel table): the table lists every kernel of the pass with its launches
    and its FETCH (doubled) / WRITE megabytes per launch - which launch over-fetches is read off it."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not on PATH", None
    tot, n, t0 = 0.0, 0, time.time()
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    perk = {}
    passes = None
    for ctr, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        d = tempfile.mkdtemp(prefix="vb_pmc_")
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "b", "--", sys.executable, os.path.abspath(__file__),
               "--streams", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-isolated", "--no-parity-check", "--no-pmc",
               "--vocoder-precision", vocoder_precision]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=max(10.0, timeout - (time.time() - t0)), env=env, cwd="/tmp")
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                return None, f"rocprofv3 --pmc {ctr} pass failed (rc {r.returncode}): {r.stderr[-200:]}", None
            m = re.search(r"PASSES_RUN=(\d+)", r.stderr)
            if m:
                passes = int(m.group(1))
            kb, launches = 0.0, 0
            for row in csv.DictReader(open(fs[0])):
                if row["Counter_Name"] != ctr:
                    continue
                kn = re.sub(r"^void ", "", row["Kernel_Name"].split("(")[0])[:72]
                e = perk.setdefault(kn, {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
                e[ctr] += mult * float(row["Counter_Value"]) * 1024.0
                if ctr == "FETCH_SIZE":
                    e["launches"] += 1
                if any(k in row["Kernel_Name"] for k in _class_patterns(cls)):
                    kb += float(row["Counter_Value"])
                    launches += 1
            tot += mult * kb * 1024.0
            n = max(n, launches)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {ctr} pass timed out", None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    table = [{"kernel": k, "launches": v["launches"], "class": next((c for c, pats in CLASS_KERNELS.items() if any(q in k for q in pats)), None),
              "fetch_mb_per_launch": v["FETCH_SIZE"] / max(v["launches"], 1) / 1e6, "write_mb_per_launch": v["WRITE_SIZE"] / max(v["launches"], 1) / 1e6,
              "total_gb": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) / 1e9}
             for k, v in sorted(perk.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])) if v["launches"]][:40]
    if passes and launches_per_pass:
        # the class's launch count of the isolated pass (the profiler's own classes: proj_in's GEMM counts as convolution work there, by
        # kernel NAME it is a GEMM) - bytes per pass of the named kernels / launches per pass of the class
        per_launch = tot / passes / launches_per_pass
        how = (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes of this command (--streams 1, {passes} passes, "
               f"{n} launches of the group's kernels = {tot / passes / 1e9:.1f} GB per pass, / {launches_per_pass:.0f} launches per pass; FETCH doubled per the gfx950 note)")
    else:
        per_launch = (tot / n) if n else None
        how = (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes of this command "
               f"(--streams 1, one pass incl. its warmup passes, {n} launches of the class; FETCH doubled per the gfx950 note)")
    return per_launch, how, table


class Telemetry:
    """shader clock / power of the AMD GPUs on this box, read from sysfs by a side thread WHILE the timed region runs (an idle
    reading says nothing).  Boxes of this pool differ by ~17 % on the clock-bound kernels (attention, band experts, GEMMs) while the
    memory-bound ones agree: the clock the timed region ran at belongs next to its throughput.  Best effort: {} when sysfs is absent."""

    def __init__(self, period=0.03):
        import glob
        import threading
        self.cards = [d for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
        self.samples, self.stop, self.period = [], False, period
        self.thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except Exception:
            return ""

    def _sample(self):
        import glob
        out = []
        for d in self.cards:
            cur = [ln for ln in self._read(os.path.join(d, "pp_dpm_sclk")).splitlines() if "*" in ln]
            mhz = int("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit())) if cur else None
            pw, cap = None, None
            for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                a, c = self._read(os.path.join(h, "power1_average")) or self._read(os.path.join(h, "power1_input")), self._read(os.path.join(h, "power1_cap"))
                pw = int(a) / 1e6 if a.strip().isdigit() else pw
                cap = int(c) / 1e6 if c.strip().isdigit() else cap
            out.append((mhz, pw, cap))
        return out

    def _run(self):
        while not self.stop:
            self.samples.append(self._sample())
            time.sleep(self.period)

    def __enter__(self):
        if self.cards:
            self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        if self.cards:
            self.thread.join(timeout=1.0)

    def summary(self, device=None):
        """this rank's card only (matched by PCI address; failing that, the card drawing the most power during the timed region)"""
        if not self.samples:
            return {}
        res = {}
        for ci, d in enumerate(self.cards):
            mhz = [s[ci][0] for s in self.samples if s[ci][0]]
            pw = [s[ci][1] for s in self.samples if s[ci][1]]
            cap = [s[ci][2] for s in self.samples if s[ci][2]]
            res[d.split("/")[4]] = {"sclk_mhz_min": min(mhz) if mhz else None, "sclk_mhz_max": max(mhz) if mhz else None,
                                    "sclk_mhz_avg": round(sum(mhz) / len(mhz)) if mhz else None,
                                    "power_w_avg": round(sum(pw) / len(pw), 1) if pw else None, "power_cap_w": cap[0] if cap else None,
                                    "samples": len(self.samples), "sysfs": os.path.realpath(d)}
        mine, how = None, None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            hit = [k for k, v in res.items() if addr in v["sysfs"].lower()]
            if len(hit) == 1:
                mine, how = hit[0], f"PCI address {addr}"
        except Exception:
            pass
        if mine is None:
            mine = max(res, key=lambda k: res[k]["power_w_avg"] or 0.0)
            how = "highest average power among the box's cards during the timed region"
        out = dict(res[mine])
        out.pop("sysfs", None)
        out.update(card=mine, matched_by=how, cards_on_box=len(res))
        return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5"])
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: 8 / 32 / 4 for c2 / c3 / c5)")
    ap.add_argument("--experts", type=int, default=None, help="Band-MoE experts per group (default 4; 8 for c3)")
    ap.add_argument("--seconds", type=float, default=None, help="clip length in seconds (default 20; 120 for c5)")
    ap.add_argument("--flow-steps", type=int, default=50)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "split"])
    ap.add_argument("--vocoder-precision", default="both", choices=["both", "fp32", "split", "fp32mf"],
                    help="VAE + vocoder arithmetic: 'fp32' = the literal 'fp32 vocoder' of configs[1] on v_mfma_f32_32x32x2_f32 (157 TFLOP/s roof); "
                         "'split' = fp32 I/O, every product as bf16 hi/lo pairs on the bf16 MFMA pipe (bf16x3, <= 3e-5 of exact fp32; priced "
                         "against bf16 peak / 3); 'fp32mf' = fp32 with F(2,3) minimal filtering on the stride-1 k = 3 / 7 / 11 layers (fp32 products, ~1.45x "
                         "fewer; conv1d_f32w.hip); 'both' (default) = the timed region runs with fp32mf (`value`: configs[1]'s fp32 vocoder) and then the same K "
                         "passes run with the direct fp32 kernels (`fp32_direct` sub-object) and with split (`split`), each with its own oracle check")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (N = 1, default c2 command, "
                                                          "rocprofv3 on PATH) that measure roofline.traffic in this run; fall back to the committed summary")
    ap.add_argument("--pmc-timeout", type=float, default=150.0)
    ap.add_argument("--scale", type=float, default=3.0)
    ap.add_argument("--streams", type=int, default=None, help="independent sub-batches per GPU, one HIP stream + host thread each")
    ap.add_argument("--split", default=None, help="explicit clips per sub-batch, e.g. 5,3 (default: --batch spread evenly over --streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true", help="skip the untimed one-stream pass that measures every kernel class alone")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--profile-timed", action="store_true", help="eager launches + HIP events on the dominant class inside the timed region")
    ap.add_argument("--cpu-flow-steps", type=int, default=4)
    ap.add_argument("--cpu-timeout", type=float, default=240.0)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="side file for the per-class / per-group / per-kernel tables (the stdout line stays <= 4 KB and names this path)")
    ap.add_argument("--save-out", default=None, help="directory: every rank writes the waveforms of its last pass (clip-indexed .npy)")
    a = ap.parse_args()
    dflt = {"c2": (8, 4, 20.0, 2), "c3": (32, 8, 20.0, 2), "c5": (4, 4, 120.0, 1)}[a.workload]
    a.batch = dflt[0] if a.batch is None else a.batch
    a.experts = dflt[1] if a.experts is None else a.experts
    a.seconds = dflt[2] if a.seconds is None else a.seconds
    a.streams = dflt[3] if a.streams is None else a.streams
    return a


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


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle on the host cores, bounded sample
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(sd_d, sd_v, sd_h, hp, flow_steps_total, cpu_steps, scale, T_lat=752, clip_seconds=20.0):
    """The CPU oracle (a port of the reference's algorithm, routed + hoisted variant) on the host cores:
    B=1, one 20 s clip; `cpu_steps` Euler steps are timed and extrapolated linearly to the full count,
    VAE decode and HiFi-GAN are timed in full."""
    from oracle import ref_cpu
    from tests.helpers import clip_batch, exp_noise
    cores = usable_cores()
    torch.set_num_threads(cores)
    inp = clip_batch(1, T_lat, L_CTX, clip0=0, seed=SEED)
    t0 = time.perf_counter()
    cc = ref_cpu.dit_precompute(sd_d, inp["t5_cond"], inp["midi"], inp["beats"], T_lat)
    cu = ref_cpu.dit_precompute(sd_d, inp["t5_uncond"], inp["midi"], inp["beats"], T_lat)
    t_pre = time.perf_counter() - t0
    noise = {(k, br): exp_noise(1, T_lat, 4, 2 * k + br, 4, seed=SEED) for k in range(cpu_steps) for br in (0, 1)}
    t0 = time.perf_counter()
    t_span, idx = ref_cpu.t_index_table(flow_steps_total + 1)
    x = inp["x_latent"].clone()
    n_run = min(cpu_steps, flow_steps_total)
    for k in range(n_run):       # bounded sample: the first n_run Euler steps of the same schedule
        ti = torch.full((1,), idx[k], dtype=torch.long)
        e_c = ref_cpu.dit_forward(sd_d, x, ti, cc, noise[(k, 0)])
        e_u = ref_cpu.dit_forward(sd_d, x, ti, cu, noise[(k, 1)])
        x = x + (t_span[k + 1] - t_span[k]) * (e_u + scale * (e_c - e_u))
    t_flow = (time.perf_counter() - t0) * (flow_steps_total / float(n_run))
    t0 = time.perf_counter()
    mel = ref_cpu.vae_decode(sd_v, x)
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
        cnd = ref_cpu.dit_precompute(sd_d, t5, inp["midi"], inp["beats"], T_lat)
        ref_cpu.dit_forward(sd_d, inp["x_latent"], ti, cnd, noise[(0, br)], dense=True)
    t_dense = (time.perf_counter() - t0) * flow_steps_total
    dense = {"value": clip_seconds / (t_dense + t_vae + t_voc), "unit": "mel-s/s",
             "what": f"reference-faithful evaluation order: dense experts + conditioning recomputed per evaluation, 1 of {flow_steps_total} "
                     f"steps timed and scaled to {t_dense:.1f}s, same VAE / HiFi-GAN times"}
    return {"value": clip_seconds / total, "unit": "mel-s/s", "cores": cores, "kind": "port", "reference_faithful": dense,
            "sample": f"B=1 x 20 s clip on {cores} threads: cond precompute {t_pre:.2f}s + {n_run} of "
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


# ------------------------------------------------------------------------------------------------------------------
def checks_fixture():
    """tests/golden/bench_c2_checks.npz (oracle/gen_bench_digest.py --checks): the oracle's replay of global clips 0 and 4 of the default
    command for passes 0 and 1, plus its states / routing indices of clip 0, pass 0 at six Euler steps"""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "bench_c2_checks.npz")
    return np.load(path) if os.path.exists(path) else None


def compare_with_oracle(z, mel, g, clip, ps):
    """one clip of one pass against the oracle's replay: latent in full (rel-L2 <= 1e-3), mel by sampled L1 (< 1e-3) and L2 digest"""
    z_ref = torch.from_numpy(g[f"z_c{clip}_p{ps}"]).double()
    z = z.detach().double().cpu()
    rel = float((z - z_ref).norm() / z_ref.norm())
    m = mel.detach().double().cpu().reshape(-1)
    pre = f"mel_c{clip}_p{ps}_"
    l1 = float((m[torch.from_numpy(g[pre + "idx"])] - torch.from_numpy(g[pre + "val"])).abs().mean())
    l2 = abs(float(m.norm()) - float(g[pre + "l2"][0])) / float(g[pre + "l2"][0])
    return {"clip": clip, "pass": ps, "ok": bool(rel <= 1e-3 and l1 < 1e-3 and l2 <= 1e-3 and bool(torch.isfinite(z).all())),
            "latent_rel_l2": rel, "mel_l1_sampled": l1, "mel_l2_rel": l2}


def routing_flips(w, g, scale_unused=None):
    """teacher-forced routing check on the bench's own engine: clip 0's row of the worker's batch is replaced by the ORACLE's state x_k
    (six Euler steps of pass 0), the network is evaluated once with the device-drawn router noise of exactly that evaluation (seed,
    global clip, step), and the routing indices of clip 0 (both gates, both CFG branches, every block) are compared with the oracle's.
    A differing index is caused by this evaluation's bf16 arithmetic alone.  -> flips per block, decisions per block."""
    eng, x0 = w["eng"], w["x0"]
    B, T = x0.shape[0], x0.shape[2]
    cond = eng.precompute_cond(w["t5"], w["midi"], w["beats"], T)
    depth = int(g["tf_routes_0"].shape[0])
    flips = [0] * depth
    decisions = 0
    for k in [int(v) for v in g["tf_steps"]]:
        x = x0.clone()
        x[0] = torch.from_numpy(g[f"tf_x_{k}"][0]).to(x.device)
        t_idx = torch.full((2 * B,), int(g[f"tf_tidx_{k}"][0]), dtype=torch.long)
        _, r = eng.forward(x, t_idx, cond, seed=SEED, clip_base=w["clip_base"], nfe=k, return_routes=True)
        r = r.cpu().numpy()                               # [depth][gate][2*B*T]: cond rows then uncond rows
        ref = g[f"tf_routes_{k}"].astype("int32")         # [depth][gate][2*T]
        mine = __import__("numpy").concatenate([r[:, :, 0:T], r[:, :, B * T:B * T + T]], axis=2)
        for i in range(depth):
            flips[i] += int((mine[i] != ref[i]).sum())
        decisions += 2 * 2 * T
    return {"per_block": flips, "decisions_per_block": decisions, "rate": sum(flips) / float(decisions * depth),
            "what": "teacher-forced on the oracle's states of clip 0, pass 0 at Euler steps " + ",".join(str(int(v)) for v in g["tf_steps"]) +
                    "; both gates x both CFG branches x 752 tokens per step and block"}


def parity_check(z0, mel0, fixture="bench_clip0.npz"):
    """clip 0 of the first timed pass (bf16 production precision, router noise drawn on the device) against the oracle's replay of
    exactly that clip (tests/golden/bench_clip0*.npz, written by oracle/gen_bench_digest.py from the CPU oracle + the host
    restatement of the device noise stream; one fixture per bench workload): north_star's tolerances, latent rel-L2 <= 1e-3 and
    mel L1 < 1e-3."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", fixture)
    if not os.path.exists(path):
        return {"ok": None, "why": f"tests/golden/{fixture} missing"}
    g = np.load(path)
    z_ref = torch.from_numpy(g["z"]).double()
    z = z0.detach().double().cpu()
    if tuple(z.shape) != tuple(z_ref.shape):
        return {"ok": None, "why": f"fixture holds a latent of shape {tuple(z_ref.shape)}, this run made {tuple(z.shape)}"}
    rel = float((z - z_ref).norm() / z_ref.norm())
    m = mel0.detach().double().cpu().reshape(-1)
    idx = torch.from_numpy(g["mel_idx"])
    l1 = float((m[idx] - torch.from_numpy(g["mel_val"])).abs().mean())
    l2 = abs(float(m.norm()) - float(g["mel_l2"][0])) / float(g["mel_l2"][0])
    return {"ok": bool(rel <= 1e-3 and l1 < 1e-3 and l2 <= 1e-3), "latent_rel_l2": rel, "mel_l1_sampled": l1, "mel_l2_rel": l2, "tol": 1e-3,
            "against": f"tests/golden/{fixture} (CPU oracle replay of clip 0, pass 0: same seed, same device-keyed router noise)"}


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
    one_device = bool(os.environ.get("VB_BENCH_ONE_DEVICE"))       # functional test of the N > 1 code path on a 1-GPU box (gloo, all ranks on cuda:0)
    if one_device:
        local = 0
    assert torch.cuda.is_available(), "bench.py measures the HIP path: no GPU visible"
    if not one_device:
        assert torch.cuda.device_count() >= world, f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()}"
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    if world > 1:
        import torch.distributed as dist
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    from tests.helpers import clip_batch
    from versband_amd import _lib as L
    from versband_amd import longform
    from versband_amd import model as vm
    from versband_amd import synth
    from versband_amd.engine import Context, DiTEngine, build_hifigan, build_vae_decoder

    long = args.workload == "c5"
    T_mel = int(round(args.seconds * 75.0))
    T_mel = (T_mel + 7) // 8 * 8                      # unit_frames_multiple (scripts/test_final.py:213,319)
    T_lat = T_mel // 2
    clip_seconds = T_mel / 75.0 if long else args.seconds
    dcfg, vcfg, hcfg = synth.DiTConfig(num_experts=args.experts), synth.VAEConfig(), synth.HifiGanConfig()
    if not long:
        assert T_lat <= dcfg.max_len, f"{args.seconds} s = {T_lat} latent frames exceeds max_len {dcfg.max_len}: use --workload c5"
    shapes = [synth.dit_shapes(dcfg), synth.vae_decoder_shapes(vcfg), synth.hifigan_shapes(hcfg)]
    sds = [synth.make_state_dict(s, SEED + i) for i, s in enumerate(shapes)] if rank == 0 else None
    bcast_ms, bcast_bytes = None, 0
    if world > 1:
        import torch.distributed as dist
        from versband_amd import dist as vdist
        torch.cuda.synchronize()
        dist.barrier()
        # the one collective of the path (product code: versband_amd/dist.py, also used by scripts/test_final.py --num_gpus N):
        # rank 0's checkpoints as ONE flat fp32 broadcast, every rank's received bytes compared
        sds, binfo = vdist.broadcast_state(sds, 0, device)
        assert binfo["checked"] and binfo["buffers"] == 1
        bcast_ms, bcast_bytes = binfo["ms"], binfo["bytes"]
        seen = torch.ones(1, device=device)
        dist.all_reduce(seen)                      # every rank reports in over the data-path backend (RCCL unless one-device test)
        assert int(seen.item()) == world
    log("weights ready; packing")
    # primary = the precision `value` is measured in: configs[1]'s fp32 VAE / vocoder - since round 6 with F(2,3) minimal filtering on the
    # layers it applies to (fp32 products on the f32 MFMA, ~1.45x fewer; held to the same bounds against the reference's outputs as the
    # direct kernels, tests/test_gpu_path.py).  Secondaries, each timed in its own region of the same K passes and verified the same way:
    # "fp32" = the direct fp32 kernels (rounds 4-5's `value`), "split" = bf16x3.
    prim = "fp32mf" if args.vocoder_precision == "both" else args.vocoder_precision
    secs = ["fp32", "split"] if args.vocoder_precision == "both" else []
    CLS = classes_for(prim)
    ctx = Context(device)
    S = max(1, args.streams)
    B = args.batch
    assert S <= B, "--streams must not exceed --batch"
    sizes = [B // S + (1 if i < B % S else 0) for i in range(S)]        # uneven splits allowed (8 clips on 3 streams: 3 + 3 + 2)
    if args.split:
        sizes = [int(v) for v in args.split.split(",")]
        assert sum(sizes) == B and all(v > 0 for v in sizes), "--split must list positive sub-batch sizes that add up to --batch"
        S = len(sizes)
    Bs = "+".join(str(n) for n in sizes) if (B % S or args.split) else str(B // S)
    idx, dts = vm.euler_tables(args.flow_steps + 1)

    def make_worker(nclips, clip_base, share=None):
        eng = DiTEngine(ctx, dcfg, sds[0], precision=args.precision, share=share)
        inp = clip_batch(nclips, T_lat, L_CTX, clip0=clip_base, seed=SEED)
        nets2 = {sec: dict(vae=build_vae_decoder(ctx, sds[1], precision=sec), voc=build_hifigan(ctx, sds[2], hcfg.as_hparams(), precision=sec),
                           z0=None, mel0=None, kept={}, wav=None) for sec in secs}
        return dict(eng=eng, vae=build_vae_decoder(ctx, sds[1], precision=prim),
                    voc=build_hifigan(ctx, sds[2], hcfg.as_hparams(), precision=prim), sec=nets2,
                    x0=inp["x_latent"].to(device), t5c=inp["t5_cond"].to(device), t5u=inp["t5_uncond"].to(device),
                    t5=torch.cat([inp["t5_cond"], inp["t5_uncond"]]).to(device), midi=inp["midi"].to(device),
                    beats=inp["beats"].to(device), stream=torch.cuda.Stream(device=device), clip_base=clip_base, wav=None, z0=None, mel0=None,
                    kept={}, nclips=nclips)

    # S independent sub-batches, each with its own engine handle (shared packed weights), on its own HIP stream driven by its own
    # host thread: kernels of different sub-batches overlap on the GPU and fill each other's tile-quantisation tails.
    workers = []
    for si in range(S):
        workers.append(make_worker(sizes[si], rank * B + sum(sizes[:si]), share=workers[0]["eng"] if workers else None))

    passes_run = [0]

    def one_pass(w, k, second=None):
        n = w["sec"][second] if second else w          # the nets and the result slots of this precision (second = a secondary's name)
        if w is workers[0]:
            passes_run[0] += 1
        if long:
            z = longform.sample_long(w["eng"], w["x0"], w["t5c"], w["t5u"], w["midi"], w["beats"], idx, dts, args.scale,
                                     window=dcfg.max_len, overlap=128, seed=SEED + k, clip_base=w["clip_base"])
            mel = n["vae"].run(z)
            n["wav"] = longform.vocode_chunked(n["voc"], mel, chunk=3000, halo=32)
        else:
            cond = w["eng"].precompute_cond(w["t5"], w["midi"], w["beats"], T_lat, persistent=True)
            z = w["eng"].sample_cfg(w["x0"], cond, idx, dts, args.scale, seed=SEED + k, clip_base=w["clip_base"])
            mel = n["vae"].run(z)
            n["wav"] = n["voc"].run(mel)
        if k == 0:
            n["z0"], n["mel0"] = z[:1], mel[:1]
        if k in keep_passes:
            # latents / mels of this pass stay resident (a few MB) and are verified AFTER the timed region: every clip the fixture covers
            # (global clips 0, 4 and 7: the first clip of each sub-batch and the last row of the batch) on the first pass, the first replay
            # and the last pass
            n["kept"][k] = (z, mel)

    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    pinned = {}

    def run_worker(w, ks, second=None):
        torch.cuda.set_device(device)           # a new host thread starts on device 0: bind it to this rank's GPU
        si = next((i for i, ww in enumerate(workers) if ww is w), 0)
        import threading
        if len(cores) >= world * S and hasattr(os, "sched_setaffinity") and threading.current_thread() is not threading.main_thread():
            # one core per (rank, stream) host thread, all distinct on the box: the launch threads of 8 ranks x 2 streams must not
            # migrate onto each other (pid 0 = the calling thread).  Worker threads only: the main thread keeps the process mask, which
            # the CPU-baseline child process inherits (pinned, it reported 1 core)
            try:
                core = cores[(local * S + si) % len(cores)] if world > 1 else cores[si % len(cores)]
                os.sched_setaffinity(0, {core})
                pinned[si] = core
            except OSError:
                pass
        with torch.cuda.stream(w["stream"]):
            for k in ks:
                one_pass(w, k, second)

    def run_passes(ks, second=None):
        import threading
        if S == 1:
            run_worker(workers[0], ks, second)
        else:
            ths = [threading.Thread(target=run_worker, args=(w, ks, second)) for w in workers]
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

    keep_passes = {0, 1, args.steps - 1}
    lib = L.load()
    torch.cuda.synchronize()
    log(f"engines built ({S} stream(s) x {Bs} clips of {clip_seconds:.1f} s, E={args.experts}); warmup")

    def read_prof(cls):
        ms, fl, by, n, nt = C.c_double(), C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
        L.check(lib.vb_prof_read(cls, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n), C.byref(nt)), "vb_prof_read")
        return ms.value, fl.value, by.value, n.value, nt.value

    for wi in range(max(args.warmup, 1)):
        run_passes([-1 - wi])
        torch.cuda.synchronize()
        for sec in secs:
            run_passes([-1 - wi], sec)
            torch.cuda.synchronize()
    if not os.environ.get("VB_NO_GRAPH") and not all(w["eng"].graphs() for w in workers):
        # the library captures the sampler loop the SECOND time it sees a call's buffers (the first call warms every kernel's
        # one-time attributes outside a capture): with --warmup 1 that capture + instantiation (tens of ms) would land in the
        # timed region - one more untimed pass takes it out.  Untimed, like the warmup; the timed region is still exactly K passes.
        run_passes([-50])
        torch.cuda.synchronize()
    assert all(torch.isfinite(w["wav"]).all() for w in workers)
    assert all(torch.isfinite(w["sec"][sec]["wav"]).all() for w in workers for sec in secs)

    # ---- every kernel class ALONE on the GPU: one stream, the whole batch of B clips, every 7th launch of each class bracketed by
    # HIP events on the launch stream.  This is the kernel-quality figure (roofline.frac): in the timed region below two
    # sub-batches share the CUs, so a launch's event-bracketed duration includes the other stream's kernels.
    # EXACT per-class totals (round 5): a class's launches are timed in rotating phases - launch i of class 0 (the GEMM class, ~1250 launches
    # per pass) in the pass with pass % 7 == i % 7, launch i of the other classes (attention 200, convolutions 129, ResBlock pairs 18 per
    # pass) in the pass with pass % 2 == i % 2 - so over PROF_PASSES = 14 passes EVERY launch of every class is timed (GEMM class twice,
    # the others seven times) and no two neighbouring launches are ever bracketed in the same pass: back-to-back event pairs double the
    # reading of the long conv kernels (60.8 ms against rocprofv3's 30), a bracketed launch between un-bracketed neighbours agrees with
    # rocprofv3 to 2 %.  ms_per_pass and the algorithmic work per launch are totals over all launches, not a sample mean scaled up
    # (rounds 3-4 timed every 7th launch: 181 GFLOP per pair launch where the true mean is 166).
    EVERY, EVERY_OTHER, PROF_PASSES = 7, 2, 14
    table, table_sec, dominant = [], {}, 0
    GROUPS = {"bf16 MFMA GEMMs of the DiT (projections, routed + band experts)": (0,), "bf16 flash attention (self + T5 cross)": (1,),
              f"VAE + vocoder convolutions in {'exact fp32 (v_mfma_f32_32x32x2_f32)' if prim == 'fp32' else 'fp32 with F(2,3) minimal filtering (v_mfma_f32_32x32x2_f32)' if prim == 'fp32mf' else 'split-bf16 (bf16x3)'}: implicit-GEMM conv1d + "
              "fused HiFi-GAN ResBlock pairs": (2, 3)}
    groups, dom_group = [], None

    def class_rows(per, cls_meta, passes):
        """per[cls] = totals over `passes` profiled passes: (ms of the timed launches, their flops, their bytes, all launches, timed launches);
        every launch is timed equally often (timed / launches-per-pass times), so totals / that count are exact per-pass figures"""
        rows = []
        for cls, (ms, fl, by, n, nt) in per.items():
            if nt == 0:
                continue
            name, bound, peak = cls_meta[cls]
            lpp = n / passes                          # launches per pass
            k = nt / lpp                              # how often each launch was timed
            # "exact" holds only if every pass launched the same kernels and the event pool never ran dry (prof_start drops timings
            # silently when it does): then lpp and k are whole numbers
            if abs(lpp - round(lpp)) > 1e-9 or abs(k - round(k)) > 1e-9:
                log(f"WARNING class '{name}': {n} launches over {passes} passes, {nt} timed - not every launch was timed equally often "
                    f"(launches/pass {lpp:.3f}, times timed {k:.3f}); its per-pass figures are estimates")
            tf, tbs = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e12
            rows.append({"class": name, "bound": bound, "launches_per_pass": lpp, "timed_launches": nt, "times_each_launch_was_timed": k,
                         "avg_launch_us": 1e3 * ms / nt, "ms_per_pass": ms / k, "algorithmic_gflop_per_launch": fl / nt / 1e9,
                         "algorithmic_gflop_per_pass": fl / k / 1e9, "algorithmic_mb_per_launch": by / nt / 1e6,
                         "tflops": tf, "frac_of_mfma_peak": tf / peak, "mfma_peak_tflops": peak, "algorithmic_tb_per_s": tbs,
                         "frac_of_hbm_peak": tbs / HBM_PEAK_TBS})
        return rows

    def profiled_passes(w, seeds0, mask, classes, secondary=None):
        """PROF_PASSES one-stream passes with the rotating phases; returns {cls: totals}"""
        tot = {c: [0.0, 0.0, 0.0, 0, 0] for c in classes}
        for i in range(PROF_PASSES):
            L.check(lib.vb_prof_enable(mask | (EVERY << 8) | (EVERY_OTHER << 16) | ((i % EVERY) << 20) | ((i % EVERY_OTHER) << 24)), "prof")
            run_worker(w, [seeds0 - i], secondary) if secondary else run_worker(w, [seeds0 - i])
            torch.cuda.synchronize()
            for c in classes:
                r = read_prof(c)
                for j in range(5):
                    tot[c][j] += r[j]
        L.check(lib.vb_prof_enable(0), "prof")
        return {c: tuple(v) for c, v in tot.items()}

    if not args.no_isolated:
        w = workers[0] if S == 1 else make_worker(B, rank * B, share=workers[0]["eng"])
        run_worker(w, [-100])
        torch.cuda.synchronize()
        per = profiled_passes(w, -101, 0xF, list(CLS))
        table = class_rows(per, CLS, PROF_PASSES)
        # "dominant" = the kernel GROUP with the largest GPU time per pass.  The two convolution classes are one group - the VAE / vocoder
        # convolutions run on two kernel families (implicit-GEMM conv1d, fused ResBlock pairs) the way the GEMM class runs on six - and
        # with the fp32 vocoder of configs[1] that group IS the largest part of a pass; every class keeps its own row in `classes`.
        def per_pass(c):      # (ms, flops, bytes, launches) of class c per pass, exact (every launch timed k times)
            ms_, fl_, by_, n_, nt_ = per[c]
            k_ = nt_ / (n_ / PROF_PASSES)
            return ms_ / k_, fl_ / k_, by_ / k_, n_ / PROF_PASSES
        for gname, ids in GROUPS.items():
            live = [c for c in ids if per[c][4]]
            tms = sum(per_pass(c)[0] for c in live)
            tfl = sum(per_pass(c)[1] for c in live)
            tby = sum(per_pass(c)[2] for c in live)
            nl = sum(per_pass(c)[3] for c in live)
            if tms > 0:
                groups.append({"group": gname, "classes": [CLS[c][0] for c in ids], "class_ids": list(ids), "ms_per_pass": tms, "launches_per_pass": nl,
                               "avg_launch_us": 1e3 * tms / nl, "tflops": tfl / (tms * 1e-3) / 1e12, "mfma_peak_tflops": CLS[ids[0]][2],
                               "frac_of_mfma_peak": tfl / (tms * 1e-3) / 1e12 / CLS[ids[0]][2], "algorithmic_mb_per_launch": tby / nl / 1e6,
                               "algorithmic_gflop_per_pass": tfl / 1e9, "ideal_ms_per_pass_at_mfma_peak": tfl / (CLS[ids[0]][2] * 1e12) * 1e3})
        dom_group = max(groups, key=lambda g: g["ms_per_pass"]) if groups else None
        dominant = max(per, key=lambda c: per_pass(c)[0] if per[c][4] else 0.0)
        if dom_group:
            dominant = max(dom_group["class_ids"], key=lambda c: per_pass(c)[0] if per[c][4] else 0.0)
        for sec in secs:
            # the convolution classes once more with each secondary's nets (classes 2 and 3 only: the DiT classes do not change)
            run_worker(w, [-110], sec)
            torch.cuda.synchronize()
            table_sec[sec] = class_rows(profiled_passes(w, -131, 0xC, [2, 3], sec), classes_for(sec), PROF_PASSES)
        if S > 1:
            del w
            torch.cuda.empty_cache()
    # The timed region runs exactly as production does - the sampler loop replayed as a captured hipGraph - so no HIP events
    # sit between its launches (a replayed graph offers none); --profile-timed restores eager launches with every 8th launch of
    # the dominant class bracketed inside the timed region (slower: the event records and ~6000 host launches per pass).
    if args.profile_timed:
        L.check(lib.vb_prof_enable((1 << dominant) | (8 << 8)), "prof")
    log(f"warmup done; dominant class = {CLS[dominant][0]}; timing {args.steps} step(s)")

    barrier()
    with Telemetry() as tele:
        t0 = time.perf_counter()
        run_passes(list(range(args.steps)))
        barrier()
        elapsed = time.perf_counter() - t0
    per_rank_ms = [1e3 * elapsed / args.steps]
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [1e3 * float(v.item()) / args.steps for v in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    log(f"timed region: {elapsed:.3f}s")
    ms, fl, by, n, nt = read_prof(dominant) if args.profile_timed else (0.0, 0.0, 0.0, 0, 0)
    L.check(lib.vb_prof_enable(0), "prof")
    elapsed_sec, per_rank_ms_sec = {}, {}
    for sec in secs:
        # the same K passes with this secondary's VAE / vocoder, timed the same way (barrier + synchronize on both sides, max over ranks)
        barrier()
        t0 = time.perf_counter()
        run_passes(list(range(args.steps)), sec)
        barrier()
        elapsed_sec[sec] = time.perf_counter() - t0
        per_rank_ms_sec[sec] = [1e3 * elapsed_sec[sec] / args.steps]
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([elapsed_sec[sec]], device=device, dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank_ms_sec[sec] = [1e3 * float(v.item()) / args.steps for v in allt]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_sec[sec] = float(t.item())
        log(f"secondary timed region ({sec} VAE / vocoder): {elapsed_sec[sec]:.3f}s")

    def verify(slots, label, with_flips):
        """everything a timed region produced that an oracle fixture covers: clip 0 of pass 0 against the workload's digest; on the default
        command also global clips 0, 4 (first clip of each sub-batch / stream) and 7 (last row of the batch: the partial-tile edge of every
        token-row launch) on the first timed pass (graph replay #1 of this seed) and on pass 1 (a replay with another seed), every kept output
        finite, and - when the last pass is neither - the last pass bitwise against an EAGER run of the same seed; with_flips: the
        teacher-forced routing-flip count per block.  `slots` = per worker, the result slots of the precision being verified."""
        fixture = None
        if not long and abs(args.seconds - 20.0) < 1e-9 and args.experts in (4, 8):
            fixture = "bench_clip0.npz" if args.experts == 4 else "bench_clip0_e8.npz"
        elif long and abs(args.seconds - 120.0) < 1e-9 and args.experts == 4 and args.batch == 4 and S == 1:
            fixture = "bench_clip0_long.npz"
        if not fixture:
            return None
        par = parity_check(slots[0]["z0"], slots[0]["mel0"], fixture)
        log(f"[{label}] parity vs oracle digest: {par}")
        g = checks_fixture() if fixture == "bench_clip0.npz" else None
        if par is None or g is None:
            return par
        clips = [int(c) for c in g["clips"]] if "clips" in g else [0, 4]
        rows, finite = [], True
        for w, n in zip(workers, slots):
            for ps, (z, mel) in sorted(n["kept"].items()):
                finite = finite and bool(torch.isfinite(z).all()) and bool(torch.isfinite(mel).all())
                if ps not in (0, 1):
                    continue
                for c in clips:
                    r = c - w["clip_base"]
                    if 0 <= r < w["nclips"]:
                        rows.append(compare_with_oracle(z[r:r + 1], mel[r:r + 1], g, c, ps))
        last = args.steps - 1
        replay_eq = None
        if last not in (0, 1) and all(w["eng"].graphs() for w in workers):
            replay_eq = True
            for w, n in zip(workers, slots):
                with torch.cuda.stream(w["stream"]):
                    cond = w["eng"].precompute_cond(w["t5"], w["midi"], w["beats"], T_lat, persistent=True)
                    z_e, _ = w["eng"].sample_cfg(w["x0"], cond, idx, dts, args.scale, seed=SEED + last, clip_base=w["clip_base"], return_traj=True)
                torch.cuda.synchronize()
                replay_eq = replay_eq and bool(torch.equal(z_e, n["kept"][last][0]))
        par["verified"] = {"clips_x_passes": rows, "all_outputs_finite": finite,
                           "last_pass": ("pass %d is covered by the oracle fixture" % last) if last in (0, 1) else
                                        {"pass": last, "graph_replay_equals_eager_bitwise": replay_eq},
                           "against": "tests/golden/bench_c2_checks.npz (oracle/gen_bench_digest.py --checks, --checks-add 7)"}
        flips_ok = True
        if with_flips:
            with torch.cuda.stream(workers[0]["stream"]):
                flips = routing_flips(workers[0], g)
            torch.cuda.synchronize()
            flips_ok = flips["rate"] <= 2e-4
            par["routing_flips"] = flips
        par["ok"] = bool(par["ok"] and finite and all(r["ok"] for r in rows) and replay_eq is not False and flips_ok)
        log(f"[{label}] verified {len(rows)} clip x pass outputs against the oracle: " + ", ".join(f"c{r['clip']}p{r['pass']} {r['latent_rel_l2']:.1e}" for r in rows) +
            f"; finite {finite}; last-pass replay == eager: {replay_eq}" +
            (f"; routing flips per block {par['routing_flips']['per_block']} of {par['routing_flips']['decisions_per_block']}" if with_flips else ""))
        return par

    parity, parity_sec = None, {}
    if rank == 0 and not args.no_parity_check and args.flow_steps == 50 and args.scale == 3.0 and args.precision == "bf16" and args.steps >= 1:
        # one oracle fixture per workload shape: 20 s clips with 4 / 8 experts (clip 0 does not depend on the batch it rides in), and
        # the 120 s long-form clip whose window rows / noise keys depend on the batch (4).  A failure anywhere fails the run (exit code 3).
        parity = verify(workers, f"{prim} VAE / vocoder", True)
        for sec in secs:
            parity_sec[sec] = verify([w["sec"][sec] for w in workers], f"{sec} VAE / vocoder", False)

    if rank == 0:
        name, bound, peak = CLS[dominant]
        conc = (fl / (ms * 1e-3) / 1e12) if (ms > 0 and nt > 0) else 0.0      # flops and time of the timed launches
        iso = next((r for r in table if r["class"] == name), None)
        achieved = iso["tflops"] if iso else conc
        pmc_cls = dominant
        if dom_group:
            # the line's roofline block describes the dominant GROUP (launch-weighted over its classes); `classes` holds every class on its own
            name, peak, achieved = dom_group["group"], dom_group["mfma_peak_tflops"], dom_group["tflops"]
            iso = dict(dom_group)
            pmc_cls = tuple(dom_group["class_ids"])
        # the committed PMC passes were taken on the default command (c2, 8 clips): for any other workload the per-launch figure does not apply
        pmc_applies = args.workload == "c2" and B == 8 and args.experts == 4 and abs(clip_seconds - 20.0) < 1e-9
        traffic, traffic_file, traffic_how, traffic_table = None, None, None, None
        if pmc_applies and world == 1 and not args.no_pmc:
            log("PMC traffic passes (rocprofv3 child runs)")
            traffic, traffic_how, traffic_table = pmc_traffic_live(pmc_cls, args.pmc_timeout, prim, iso["launches_per_pass"] if iso else None)
            log(f"traffic: {traffic} ({traffic_how})")
        if traffic is None and pmc_applies:
            live_why = traffic_how
            traffic, traffic_file = pmc_traffic(pmc_cls)
            traffic_how = (f"file: {traffic_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH doubled per the gfx950 note"
                           + (f"; live measurement unavailable: {live_why}" if live_why else "") + ")") if traffic else live_why
        total_mel_s = world * B * clip_seconds * args.steps
        wl = {"c2": f"{B} x {clip_seconds:.0f} s clips per GPU (T_lat={T_lat}, T_mel={T_mel}, 24 kHz)",
              "c3": f"Band-MoE stress: {B} x {clip_seconds:.0f} s clips per GPU, num_experts={args.experts} ({2 * B * T_lat} token rows per evaluation)",
              "c5": f"long-form: {B} x {clip_seconds:.0f} s clips per GPU (T_lat={T_lat} in windows of {dcfg.max_len} tokens, overlap 128, "
                    f"cross-faded; VAE on the whole latent; vocoder in 3000-frame chunks with 32-frame halos)"}[args.workload]
        out = {
            "metric": "mel-seconds generated/sec (%d s clip, %d flow steps)" % (round(clip_seconds), args.flow_steps),
            "value": total_mel_s / elapsed,
            "unit": "mel-s/s",
            "n_gpus": world,
            "ranks": {"world": world, "backend": ("gloo (VB_BENCH_ONE_DEVICE functional test)" if one_device else "nccl (RCCL)") if world > 1 else None,
                      "weight_broadcast_ms": bcast_ms, "weight_broadcast_bytes": bcast_bytes,
                      "weight_broadcast_gbps": (bcast_bytes / (bcast_ms * 1e-3) / 1e9) if bcast_ms else None,
                      "per_rank_ms": per_rank_ms, "host_thread_cores_rank0": pinned, "collectives_in_timed_region": 0},
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("bf16 DiT (fp32 accumulate)" if args.precision == "bf16" else "bf16x3 split DiT") +
                     (" + fp32-I/O VAE/vocoder on split-bf16 (bf16x3) MFMA, <=3e-5 of exact fp32" if prim == "split" else
                      " + fp32 VAE/vocoder (v_mfma_f32_32x32x2_f32, fp32 products; F(2,3) minimal filtering on the stride-1 k = 3 / 7 / 11 layers)"
                      if prim == "fp32mf" else " + fp32 VAE/vocoder (v_mfma_f32_32x32x2_f32, exact fp32 products)"),
            "data": "synthetic (seeded PRNG clips, random-init checkpoints of the configured architecture)",
            "config": {"workload": wl + f", {args.flow_steps} Euler steps x 2 NFE (CFG scale {args.scale}), Band-MoE E={args.experts}, "
                                        "VAE decode + HiFi-GAN V1-like (8*5*4*2), configs/vocal2music.yaml; VAE / vocoder arithmetic: " +
                                        ("split-bf16 (bf16x3 products, fp32 I/O and accumulation, <= 3e-5 of exact fp32; `--vocoder-precision fp32` "
                                         "runs the literal fp32 kernels)" if prim == "split" else
                                         "fp32 on the f32 MFMA with F(2,3) minimal filtering (fp32 products, ~1.45x fewer on the layers it applies to; "
                                         "<= 2e-6 of the direct fp32 kernels)" + ("; the same K passes with the direct fp32 kernels and with the bf16x3 "
                                         "VAE / vocoder are timed in their own regions: `fp32_direct`, `split`" if secs else "") if prim == "fp32mf" else
                                         "exact fp32 (f32 MFMA, configs[1] as written)" + ("; the same K passes with the bf16x3 VAE / vocoder are timed in a "
                                                                                            "second region and reported under `split`" if secs else "")),
                       "vocoder_precision": prim,
                       "baseline_config": {"c2": "configs[1]", "c3": "configs[2]", "c5": "configs[4]"}[args.workload],
                       "clips_per_gpu": B, "clip_seconds": clip_seconds, "flow_steps": args.flow_steps, "precision": args.precision, "experts": args.experts,
                       "sampler_loop": ("hipGraph replay (%d graph(s) on rank 0)" % sum(w["eng"].graphs() for w in workers))
                       if any(w["eng"].graphs() for w in workers) else "eager launches",
                       "streams_per_gpu": S, "parallelism": f"batch-shard x{world} ({S} concurrent sub-batches of {Bs} clips per GPU)"},
            "parity_check": parity,
            "device": {"name": torch.cuda.get_device_name(device), "clocks_during_timed_region": tele.summary(device)},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "how": ("dominant group = largest GPU time per pass (the two convolution classes count as one group, like the six GEMM kernels "
                                 "count as one class); achieved = algorithmic flops of its launches / their durations, from the event-timed launches "
                                 "of each of its classes scaled to the class's launch count, measured in this run with every class ALONE on the GPU "
                                 "(one stream, whole batch, launches bracketed by HIP events on the launch stream in rotating phases over "
                                 f"{PROF_PASSES} passes so that EVERY launch is timed - the GEMM class twice, the others seven times - and never "
                                 "two neighbours in one pass: totals, not a scaled sample); `classes` and `groups` list every class / group the "
                                 "same way") if iso else "timed-region launches (no isolated pass)",
                         "avg_launch_us": iso["avg_launch_us"] if iso else ((1e3 * ms / nt) if nt else None),
                         "traffic": traffic, "traffic_per_kernel": traffic_table,
                         "traffic_source": traffic_how if pmc_applies else
                                           "none for this workload: the PMC passes cover the default c2 command only",
                         "algorithmic_bytes_per_launch": (iso["algorithmic_mb_per_launch"] * 1e6) if iso else None,
                         "timed_region": ({"achieved": conc, "frac": conc / peak, "avg_launch_us": (1e3 * ms / nt) if nt else None,
                                           "launches_per_step": n / max(args.steps, 1), "timed_launches": nt,
                                           "note": "every 8th launch of the dominant class bracketed by HIP events INSIDE the timed region; with "
                                                   "streams_per_gpu > 1 a launch shares the GPU with the other sub-batch's kernels, so its "
                                                   "duration includes that overlap (not a kernel-quality figure)"} if args.profile_timed else
                                          "not event-bracketed: the timed region replays the sampler loop as a hipGraph (run with --profile-timed "
                                          "for eager launches + in-region events)"),
                         "classes": table, "groups": [{k: v for k, v in g.items() if k != "class_ids"} for g in groups]},
        }
        if groups:
            # the whole path against its roofs: every group's algorithmic flops per pass at the dense MFMA peak of its arithmetic (bf16 2.5 PF,
            # f32 157.3 TF), summed, over the measured time of a pass - the number a round should move (round 4: 67.9 ms ideal / 143.4 = 0.47)
            ideal = sum(g["ideal_ms_per_pass_at_mfma_peak"] for g in groups)
            out["roofline"]["path_roofline"] = {
                "ideal_ms_per_pass": ideal, "ms_per_step": 1e3 * elapsed / args.steps, "frac": ideal / (1e3 * elapsed / args.steps),
                "algorithmic_tflop_per_pass": sum(g["algorithmic_gflop_per_pass"] for g in groups) / 1e3,
                "per_group": [{"group": g["group"], "algorithmic_gflop_per_pass": g["algorithmic_gflop_per_pass"], "peak_tflops": g["mfma_peak_tflops"],
                               "ideal_ms": g["ideal_ms_per_pass_at_mfma_peak"], "measured_ms_alone": g["ms_per_pass"]} for g in groups],
                "how": "sum over kernel groups of (algorithmic flops per pass / dense MFMA peak of the group's arithmetic) / ms_per_step; flops = the "
                       "library's per-launch algorithmic counts (DESIGN.md section 4) summed over every launch of a pass"}
        what = {"split": "the same workload and the same K passes with the VAE / vocoder on split-bf16 (bf16x3: every product as hi*hi + lo*hi + hi*lo on "
                         "the bf16 MFMA pipe, fp32 I/O and accumulation, <= 3e-5 of exact fp32) - narrower than the fp32 arithmetic configs[1] names, "
                         "so it is reported beside `value`, not as it",
                "fp32": "the same workload and the same K passes with the DIRECT fp32 convolution kernels everywhere (no minimal filtering; the 64-channel "
                        "ResBlock pairs fused): what `value` measured in rounds 4-5"}
        for sec in secs:
            out["fp32_direct" if sec == "fp32" else sec] = {
                "what": what[sec], "value": total_mel_s / elapsed_sec[sec], "unit": "mel-s/s", "ms_per_step": 1e3 * elapsed_sec[sec] / args.steps,
                "steps": args.steps, "per_rank_ms": per_rank_ms_sec[sec], "vocoder_precision": sec, "parity_check": parity_sec.get(sec),
                "classes": table_sec.get(sec, [])}
        if prim == "fp32mf" and dom_group and tuple(dom_group.get("class_ids", ())) == (2, 3) and table_sec.get("fp32"):
            # direct-convolution-equivalent throughput of the dominant group: the flops the direct kernels execute for the same layers (the
            # `fp32` secondary's class table) over this run's group time; `frac` above stays on the flops the minimal-filtering kernels execute
            dgf = sum(r["algorithmic_gflop_per_pass"] for r in table_sec["fp32"])
            out["roofline"]["direct_equivalent"] = {"achieved": dgf / dom_group["ms_per_pass"], "unit": "TFLOP/s",
                                                    "frac": dgf / dom_group["ms_per_pass"] / dom_group["mfma_peak_tflops"],
                                                    "direct_gflop_per_pass": dgf, "executed_gflop_per_pass": dom_group["algorithmic_gflop_per_pass"]}
            pr_ = out["roofline"].get("path_roofline")
            if pr_:
                # the same for the whole path: the convolution group's ideal time from the DIRECT convolution's flops (comparable with rounds 4-5)
                ideal_d = pr_["ideal_ms_per_pass"] - dom_group["ideal_ms_per_pass_at_mfma_peak"] + dgf / dom_group["mfma_peak_tflops"]
                pr_["direct_equivalent_ideal_ms_per_pass"] = ideal_d
                pr_["direct_equivalent_frac"] = ideal_d / pr_["ms_per_step"]
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            log("cpu baseline (subprocess, bounded)")
            out["cpu_baseline"] = cpu_baseline_subprocess(args)
        detail_path = args.detail
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)) or ".", exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:                     # a read-only checkout must not cost the line
            log(f"could not write {detail_path}: {e}")
            detail_path = None
        sys.stderr.flush()
        print(compact_line(out, detail_path), flush=True)
        if (parity is not None and parity.get("ok") is False) or any(v is not None and v.get("ok") is False for v in parity_sec.values()):
            log("PARITY CHECK FAILED")
    if rank == 0:
        log(f"PASSES_RUN={passes_run[0]}")      # (read back by the PMC child-run parser: counter sums -> bytes per pass)
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
    if rank == 0 and ((parity is not None and parity.get("ok") is False) or any(v is not None and v.get("ok") is False for v in parity_sec.values())):
        sys.exit(3)


if __name__ == "__main__":
    main()
