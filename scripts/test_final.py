#!/usr/bin/env python3
"""AccompBand inference entry point on the MI355X-native engine.

Keeps the CLI and the semantics of the reference's scripts/test_final.py (flags :34-98, per-item loop
:376-457, output naming :429-457, clap.csv :462) without its defects: `--ddim_steps`
really sets the number of Euler steps, `x_T` really is the start latent, no hard-coded paths.
One process per GPU (`--num_gpus`, items sharded rank::world like DistributedSampler); no collective.

`--synthetic N` runs N seeded synthetic items with random-init checkpoints (no dataset / checkpoint needed).
"""
from __future__ import annotations

import argparse
import csv
import os
import sys
import wave
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ldm.models.diffusion.cfm1_audio_sampler import CFMSampler  # noqa: E402
from ldm.util import instantiate_from_config  # noqa: E402
from versband_amd import synth  # noqa: E402
from versband_amd.model import load_config, normalize_loudness  # noqa: E402
from vocoder.hifigan import HifiGAN  # noqa: E402

UNIT_FRAMES_MULTIPLE = 8        # test_final.py:213
MEL_DOWNSAMPLE = 2              # latent length = int(T_mel / 2)  (:389)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--config", type=str, default=os.path.join(ROOT, "configs", "vocal2music.yaml"))
    p.add_argument("--ckpt", type=str, default=None)
    p.add_argument("--vocoder_ckpt", type=str, default=None)
    p.add_argument("--manifest_path", type=str, default=None)
    p.add_argument("--other_condition", type=str, default=None)
    p.add_argument("--ddim_steps", type=int, default=24, help="number of Euler flow steps (the reference always ran 24)")
    p.add_argument("--n_samples", type=int, default=1)
    p.add_argument("--scale", type=float, default=3.0)
    p.add_argument("--scales", type=str, default="1-3")
    p.add_argument("--save_dir", type=str, default="test")
    p.add_argument("--save_plot", action="store_true")
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--sample_rate", type=int, default=24000)
    p.add_argument("--synthetic", type=int, default=0, help="run N synthetic items with random-init checkpoints")
    p.add_argument("--synthetic_frames", type=int, default=1500)
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "split"])
    p.add_argument("--seed", type=int, default=1234)
    return p.parse_args()


def pad_or_cut_xd(x: np.ndarray, length: int, dim: int, pad_value=0):
    """the helper the reference imports but never defines."""
    n = x.shape[dim]
    if n >= length:
        return np.take(x, np.arange(length), axis=dim)
    pad = [(0, 0)] * x.ndim
    pad[dim] = (0, length - n)
    return np.pad(x, pad, constant_values=pad_value)


def load_samples_from_tsv(tsv_path):
    with open(tsv_path) as f:
        reader = csv.DictReader(f, delimiter="\t", quotechar=None, doublequote=False, lineterminator="\n", quoting=csv.QUOTE_NONE)
        return [dict(e) for e in reader]


class InferDataset:
    """Intended behaviour of test_final.py:196-340: item -> caption, midi/beats [1,T], length rounded up to 8 frames."""

    def __init__(self, manifest_path, other_condition):
        self.items = load_samples_from_tsv(manifest_path)
        self.other = np.load(other_condition, allow_pickle=True).item() if other_condition else {}

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        it = self.items[i]
        name = it.get("name", str(i))
        extra = self.other.get(name, {})
        midi = np.asarray(extra.get("midi", np.zeros(1, dtype=np.int64))).reshape(1, -1)
        beats = np.asarray(extra.get("beats", np.zeros(midi.shape[1], dtype=np.int64))).reshape(1, -1)
        T = int(np.ceil(midi.shape[1] / UNIT_FRAMES_MULTIPLE) * UNIT_FRAMES_MULTIPLE)
        midi, beats = pad_or_cut_xd(midi, T, 1, 128), pad_or_cut_xd(beats, T, 1, 2)
        style = it.get("caption", "").split("<psep>")[0]
        return {"name": name, "caption": f"Style: {style} Musical: ", "midi": torch.from_numpy(midi).long(),
                "beats": torch.from_numpy(beats).long(), "acoustic": torch.zeros(20, T), "audio_path": it.get("audio_path")}


class SyntheticDataset:
    def __init__(self, n, frames, seed):
        self.n, self.frames, self.seed = n, frames, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        T_mel = int(np.ceil(self.frames / UNIT_FRAMES_MULTIPLE) * UNIT_FRAMES_MULTIPLE)
        c = synth.make_clip_inputs(self.seed, i, T_mel // 2, valid_mel=self.frames)
        return {"name": f"synthetic{i:04d}", "caption": c["t5_cond"], "uncond_caption": c["t5_uncond"], "midi": c["midi"],
                "beats": c["beats"], "acoustic": torch.zeros(20, T_mel), "audio_path": None, "clip": i}


def safe_path(path):
    os.makedirs(Path(path).parent, exist_ok=True)
    return path


def write_wav_pcm16(path, wav, sr):
    try:
        import soundfile as sf
        sf.write(safe_path(path), wav, sr, subtype="PCM_16")
        return
    except ImportError:
        pass
    pcm = (np.clip(wav, -1.0, 1.0) * 32767.0).astype("<i2")
    with wave.open(safe_path(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(pcm.tobytes())


def initialize_model(args, device):
    config = load_config(args.config)
    config.model.params["precision"] = args.precision
    model = instantiate_from_config(config.model)
    if args.ckpt:
        sd = torch.load(args.ckpt, map_location="cpu")["state_dict"]
    else:       # random-init checkpoint of the configured architecture
        dcfg = model.model.diffusion_model.cfg
        sd = {"model.diffusion_model." + k: v for k, v in synth.make_state_dict(synth.dit_shapes(dcfg), args.seed).items()}
        sd.update({"first_stage_model." + k: v for k, v in
                   synth.make_state_dict(synth.vae_decoder_shapes(synth.VAEConfig()), args.seed + 1).items()})
    model.load_state_dict(sd, strict=False)
    model = model.to(device)
    return CFMSampler(model, num_timesteps=1000)


def make_vocoder(args, device, tmp_dir):
    if args.vocoder_ckpt:
        return HifiGAN(vocoder_ckpt=args.vocoder_ckpt, device=device)
    import yaml
    hcfg = synth.HifiGanConfig()
    os.makedirs(tmp_dir, exist_ok=True)
    yaml.safe_dump(hcfg.as_hparams(), open(os.path.join(tmp_dir, "config.yaml"), "w"))
    torch.save({"state_dict": {"model_gen": synth.make_state_dict(synth.hifigan_shapes(hcfg), args.seed + 2)}},
               os.path.join(tmp_dir, "model_ckpt_steps_0.ckpt"))
    return HifiGAN(vocoder_ckpt=tmp_dir, device=device)


@torch.no_grad()
def gen_song(rank, args):
    device = torch.device(f"cuda:{int(rank)}")
    dataset = SyntheticDataset(args.synthetic, args.synthetic_frames, args.seed) if args.synthetic else \
        InferDataset(args.manifest_path, args.other_condition)
    indices = list(range(len(dataset)))[rank::args.num_gpus]          # DistributedSampler(shuffle=False) sharding
    sampler = initialize_model(args, device)
    vocoder = make_vocoder(args, device, os.path.join(args.save_dir, f".synthetic_vocoder_{rank}"))
    scales = [float(s) for s in args.scales.split("-")] if args.scales else [args.scale]
    rows = []
    for item_idx, gi in enumerate(indices):
        item = dataset[gi]
        midi, beats, acoustic = item["midi"].to(device), item["beats"].to(device), item["acoustic"].to(device)
        n = args.n_samples
        for scale in scales:
            latent_length = int(acoustic.shape[1] / MEL_DOWNSAMPLE)
            embed_dim = sampler.model.first_stage_model.embed_dim
            start_code = torch.randn(n, embed_dim, latent_length, generator=torch.Generator().manual_seed(args.seed + gi)).to(device)
            cap = item["caption"]
            cond_in = {"caption": torch.stack([cap] * n) if torch.is_tensor(cap) else [cap] * n,
                       "acoustic": {"acoustic": torch.stack([acoustic] * n), "midi": torch.stack([midi] * n).long(),
                                    "beats": torch.stack([beats] * n).long()}, "name": [item["name"]] * n}
            c = sampler.model.get_learned_conditioning(cond_in)
            uc = None
            if scale != 1.0:
                ucap = item.get("uncond_caption", "")
                uc_in = dict(cond_in)
                uc_in["caption"] = torch.stack([ucap] * n) if torch.is_tensor(ucap) else [ucap] * n
                uc = sampler.model.get_learned_conditioning(uc_in)
            z, _ = sampler.sample_cfg(S=args.ddim_steps, cond=c, batch_size=n, shape=[embed_dim, latent_length], verbose=False,
                                      unconditional_guidance_scale=scale, unconditional_conditioning=uc, x_T=start_code,
                                      x_latent=start_code, timesteps=args.ddim_steps + 1, seed=args.seed, clip_base=gi * n)
            mel = sampler.model.decode_first_stage(z)
            out_dir = os.path.join(args.save_dir, f"cond_gtcodec_accomp_scale_{scale}")
            for k, spec in enumerate(mel):
                wav = vocoder(spec.transpose(0, 1).cpu())
                wav = normalize_loudness(wav, -23)
                path = os.path.join(out_dir, f"{rank}-{item_idx:04d}[{k}][accomp].wav")
                write_wav_pcm16(path, wav, args.sample_rate)
                rows.append({"audio_path": path, "caption": cap if isinstance(cap, str) else item["name"], "name": item["name"]})
    csv_path = safe_path(os.path.join(args.save_dir, f"clap.csv" if args.num_gpus == 1 else f"clap.{rank}.csv"))
    with open(csv_path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["audio_path", "caption", "name"], delimiter="\t")
        w.writeheader()
        w.writerows(rows)
    print(f"[rank {rank}] wrote {len(rows)} wav files, {csv_path}")


if __name__ == "__main__":
    args = parse_args()
    if args.num_gpus > 1:
        import torch.multiprocessing as mp
        mp.spawn(gen_song, nprocs=args.num_gpus, args=(args,))
    else:
        gen_song(0, args=args)
