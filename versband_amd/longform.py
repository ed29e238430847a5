"""Long-form generation (BASELINE config 5: 120 s accompaniment) - build-defined, the reference has none.

The reference cannot go past max_len = 1500 latent tokens (40 s): its RoPE table asserts
(vocal2music_moe.py:421, flag_large_dit_moe.py:232) and no chunking exists anywhere in the tree
(SURVEY Q14).  Here a long clip is cut into windows of <= max_len latent tokens that overlap by
`overlap` tokens; every window is an independent "clip" for the sampler (own slice of the midi/beats
tracks, same caption, own slice of the start noise), so all windows of all clips run as ONE batch
through vb_sample_cfg.  Window latents are cross-faded linearly over the overlaps.  The VAE decoder is
applied to the whole latent (it is length-agnostic), the HiFi-GAN - fully convolutional - is run in
chunks with a halo larger than its receptive field and the chunks are stitched (overlap-discard), which
reproduces whole-clip vocoding exactly.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

Tensor = torch.Tensor


def plan_windows(T: int, window: int, overlap: int) -> List[Tuple[int, int]]:
    """[(start, length)] covering [0,T) with windows of `window` tokens overlapping by `overlap`."""
    if T <= window:
        return [(0, T)]
    assert 0 <= overlap < window
    hop = window - overlap
    out, s = [], 0
    while True:
        if s + window >= T:
            out.append((T - window, window))
            break
        out.append((s, window))
        s += hop
    return out


def crossfade_windows_hip(lib, zw: Tensor, plan: Sequence[Tuple[int, int]], B: int, T: int) -> Tensor:
    """the product path of crossfade_windows: window results zw [nw*B, C, n] (row = w*B + b) -> [B, C, T] by vb_crossfade_windows (one kernel,
    same window order and arithmetic as the torch restatement below, which the oracle fixtures were generated with)"""
    import ctypes as C
    from . import _lib as L
    nw, n = len(plan), plan[0][1]
    assert all(m == n for _, m in plan) and zw.shape[0] == nw * B and zw.shape[2] == n
    zw = zw.contiguous()
    out = torch.empty(B, zw.shape[1], T, dtype=torch.float32, device=zw.device)
    starts = (C.c_int32 * nw)(*[s for s, _ in plan])
    L.check(lib.vb_crossfade_windows(L.ptr(zw), starts, nw, B, zw.shape[1], n, T, L.ptr(out), L.stream_ptr()), "vb_crossfade_windows")
    return out


def crossfade_windows(parts: Sequence[Tensor], plan: Sequence[Tuple[int, int]], T: int) -> Tensor:
    """Blend window results [B,C,len] into [B,C,T]: linear ramps over every overlap, weights sum to one.  (Torch restatement: the CPU
    oracle's digest generator uses it - oracle/gen_bench_digest.py --long; the product path is crossfade_windows_hip.)"""
    B, C = parts[0].shape[:2]
    acc = torch.zeros(B, C, T, dtype=parts[0].dtype, device=parts[0].device)
    wsum = torch.zeros(T, dtype=parts[0].dtype, device=parts[0].device)
    for i, ((s, n), p) in enumerate(zip(plan, parts)):
        w = torch.ones(n, dtype=p.dtype, device=p.device)
        if i > 0:
            ov = plan[i - 1][0] + plan[i - 1][1] - s
            if ov > 0:
                w[:ov] = torch.linspace(0, 1, ov + 2, dtype=p.dtype, device=p.device)[1:-1]
        if i + 1 < len(plan):
            ov = s + n - plan[i + 1][0]
            if ov > 0:
                w[n - ov:] = torch.minimum(w[n - ov:], torch.linspace(1, 0, ov + 2, dtype=p.dtype, device=p.device)[1:-1])
        acc[:, :, s:s + n] += p * w
        wsum[s:s + n] += w
    return acc / wsum


def sample_long(engine, x0: Tensor, t5_cond: Tensor, t5_uncond: Tensor, midi: Tensor, beats: Tensor, t_idx_table, dt_table,
                scale: float, window: int = 1500, overlap: int = 128, seed: int = 0, clip_base: int = 0) -> Tensor:
    """x0 [B,C,T], t5_* [B,L,1024], midi/beats [B,1,2T] -> z [B,C,T] for T beyond the DiT's max_len."""
    B, C, T = x0.shape
    window = min(window, engine.cfg.max_len)
    plan = plan_windows(T, window, overlap)
    nw = len(plan)
    n = plan[0][1]
    midi, beats = midi.reshape(B, -1), beats.reshape(B, -1)
    if midi.shape[1] != beats.shape[1] or abs(midi.shape[1] - 2 * T) > 4:
        raise ValueError(f"midi/beats tracks of {midi.shape[1]} / {beats.shape[1]} frames do not match 2 x {T} latent frames")
    if midi.shape[1] < 2 * T:         # a slightly shorter track: repeat the last frame, as the stem does for |T - T_mel/2| <= 2
        pad = 2 * T - midi.shape[1]
        midi = torch.cat([midi, midi[:, -1:].expand(B, pad)], dim=1)
        beats = torch.cat([beats, beats[:, -1:].expand(B, pad)], dim=1)
    # windows become extra batch rows: row = w * B + b
    xw = torch.cat([x0[:, :, s:s + n] for s, _ in plan], dim=0)
    mw = torch.cat([midi[:, 2 * s:2 * (s + n)] for s, _ in plan], dim=0)
    bw = torch.cat([beats[:, 2 * s:2 * (s + n)] for s, _ in plan], dim=0)
    t5 = torch.cat([t5_cond.repeat(nw, 1, 1), t5_uncond.repeat(nw, 1, 1)], dim=0)
    cond = engine.precompute_cond(t5, mw, bw, n, persistent=True)
    zw = engine.sample_cfg(xw, cond, t_idx_table, dt_table, scale, seed=seed, clip_base=clip_base * nw)
    if nw == 1:
        return zw
    if zw.is_cuda and all(m == n for _, m in plan):
        return crossfade_windows_hip(engine.ctx.lib, zw, plan, B, T)
    parts = [zw[i * B:(i + 1) * B] for i in range(nw)]
    return crossfade_windows(parts, plan, T)


def vocode_chunked(vocoder_net, mel: Tensor, chunk: int = 2048, halo: int = 32) -> Tensor:
    """mel [B,80,T] -> wav [B,1,T*hop] in chunks of `chunk` frames with `halo` frames of context on both sides;
    identical to whole-clip vocoding when halo >= the generator's receptive field (in frames)."""
    B, _, T = mel.shape
    hop = vocoder_net.out_tmul
    if T <= chunk + 2 * halo:
        return vocoder_net.run(mel)
    if hasattr(vocoder_net, "run_chunked"):
        return vocoder_net.run_chunked(mel, chunk, halo)      # the product path: loop, slicing and stitching inside the library
    out = torch.empty(B, vocoder_net.out_ch, T * hop, dtype=torch.float32, device=mel.device)
    s = 0
    while s < T:
        e = min(T, s + chunk)
        lo, hi = max(0, s - halo), min(T, e + halo)
        w = vocoder_net.run(mel[:, :, lo:hi].contiguous())
        out[:, :, s * hop:e * hop] = w[:, :, (s - lo) * hop:(e - lo) * hop]
        s = e
    return out
