"""Log-mel front-end on the HIP library - host mirror of the reference's `MelNet` (preprocess/NAT_mel.py:42-86).

    MelNet(hparams, device)(wav [B, L] or [L])  ->  log10-mel [B, n_mels, L // hop]

The reference builds its filterbank with `librosa.filters.mel` (librosa==0.10.1 in its requirements.txt:3; the package is not in
this image).  `mel_filterbank` restates that function's published algorithm for its defaults (Slaney mel scale, `norm='slaney'`):
nothing here can pin it against librosa itself - see DESIGN.md 7 (N4) - while the STFT / magnitude / log part is pinned against the
reference's own MelNet (tests/golden/melnet.npz, generated with this filterbank injected for the missing import).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L
from .engine import Context

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------------------------
# host-side constants (numpy, float64 -> float32)
# ------------------------------------------------------------------------------------------------------------------
_F_SP = 200.0 / 3.0             # Slaney scale: linear below 1 kHz (66.67 Hz per mel) ...
_BREAK_HZ = 1000.0
_BREAK_MEL = _BREAK_HZ / _F_SP
_LOGSTEP = math.log(6.4) / 27.0  # ... and 27 mels per factor 6.4 above


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / _F_SP
    return np.where(f >= _BREAK_HZ, _BREAK_MEL + np.log(np.maximum(f, _BREAK_HZ) / _BREAK_HZ) / _LOGSTEP, lin)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= _BREAK_MEL, _BREAK_HZ * np.exp(_LOGSTEP * (m - _BREAK_MEL)), _F_SP * m)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: Optional[float] = None) -> np.ndarray:
    """[n_mels, n_fft//2 + 1] float32 triangular filters, area-normalised (what `librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=,
    fmax=)` documents for htk=False, norm='slaney'): band edges equally spaced on the Slaney mel scale, each triangle rising from
    edge i to edge i+1 and falling to edge i+2, scaled by 2 / (edge[i+2] - edge[i])."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    bins = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    dist = edges[:, None] - bins[None, :]
    fb = np.zeros((n_mels, bins.size), dtype=np.float32)
    for i in range(n_mels):
        rise = -dist[i] / width[i]
        fall = dist[i + 2] / width[i + 1]
        fb[i] = np.maximum(0.0, np.minimum(rise, fall))
    fb *= (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, None]
    return fb


def hann_window(win: int, n_fft: int) -> np.ndarray:
    """torch.hann_window(win) (periodic), centred and zero-padded to n_fft the way torch.stft does for win_length < n_fft."""
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win, dtype=np.float64) / win)
    out = np.zeros(n_fft, dtype=np.float64)
    left = (n_fft - win) // 2
    out[left:left + win] = w.astype(np.float32)       # the reference's window is a float32 tensor
    return out


def im_offset(n_fft: int) -> int:
    return (n_fft // 2 + 1 + 3) // 4 * 4


def dft_weights(n_fft: int, hop: int, win: int) -> np.ndarray:
    """[n_fft/hop][hop][Co4] float32: the windowed one-sided DFT basis laid out as convolution taps over hop-blocks
    (include/versband_hip.h, vb_melnet_load)."""
    assert n_fft % hop == 0 and win <= n_fft
    nb, off = n_fft // 2 + 1, im_offset(n_fft)
    n = np.arange(n_fft, dtype=np.int64)
    f = np.arange(nb, dtype=np.int64)
    ang = 2.0 * np.pi * ((n[:, None] * f[None, :]) % n_fft).astype(np.float64) / n_fft     # exact argument reduction
    w = hann_window(win, n_fft)[:, None]
    out = np.zeros((n_fft, 2 * off), dtype=np.float32)
    out[:, :nb] = w * np.cos(ang)
    out[:, off:off + nb] = -w * np.sin(ang)
    return out.reshape(n_fft // hop, hop, 2 * off)


# ------------------------------------------------------------------------------------------------------------------
# the operator
# ------------------------------------------------------------------------------------------------------------------
class MelNet(torch.nn.Module):
    """Same constructor / call surface as preprocess/NAT_mel.py:42-86: `MelNet(hparams, device)`, `.to(device)`,
    `forward(y, center=False, complex=False)`.  hparams keys: fft_size, audio_num_mel_bins, audio_sample_rate, hop_size, win_size,
    fmin, fmax.  There is no CPU path: the module needs a GPU device before it is called."""

    def __init__(self, hparams: Dict, device="cpu") -> None:
        super().__init__()
        self.n_fft = int(hparams["fft_size"])
        self.num_mels = int(hparams["audio_num_mel_bins"])
        self.sampling_rate = int(hparams["audio_sample_rate"])
        self.hop_size = int(hparams["hop_size"])
        self.win_size = int(hparams["win_size"])
        self.fmin = hparams["fmin"]
        self.fmax = hparams["fmax"]
        if self.n_fft % self.hop_size or self.n_fft % 2 or (self.n_fft - self.hop_size) % 2:
            raise ValueError(f"MelNet: fft_size {self.n_fft} must be an even multiple of hop_size {self.hop_size}")
        self._basis = mel_filterbank(self.sampling_rate, self.n_fft, self.num_mels, self.fmin, self.fmax)
        self._dft = dft_weights(self.n_fft, self.hop_size, self.win_size)
        self.mel_basis = torch.from_numpy(self._basis)
        self.hann_window = torch.hann_window(self.win_size)
        self.device = torch.device("cpu")
        self._ctx = None
        self._ws = None
        if torch.device(device).type != "cpu":
            self.to(device)

    def to(self, device, **kwargs):
        device = torch.device(device)
        self.mel_basis = self.mel_basis.to(device)
        self.hann_window = self.hann_window.to(device)
        self.device = device
        if device.type == "cuda":
            self._ctx = Context(device)
            dev = self._ctx.device
            self._w = torch.from_numpy(self._dft).to(dev).contiguous()
            self._bt = torch.from_numpy(np.ascontiguousarray(self._basis.T)).to(dev).contiguous()
            self._cfg = L.MelConfig(self.n_fft, self.hop_size, self.num_mels)
            L.check(self._ctx.lib.vb_melnet_load(self._ctx.handle, C.byref(self._cfg), L.ptr(self._w), L.ptr(self._bt)), "vb_melnet_load")
            self._ws = None
        return self

    def frames(self, n_samples: int, center: bool = False) -> int:
        pad = (self.n_fft - self.hop_size) // 2 + (self.n_fft // 2 if center else 0)
        return max(0, 1 + (n_samples + 2 * pad - self.n_fft) // self.hop_size)

    @torch.no_grad()
    def forward(self, y, center: bool = False, complex: bool = False):
        if self._ctx is None:
            raise L.VersbandError("MelNet runs on the HIP library only: move it to a GPU device first (.to('cuda:0'))")
        if isinstance(y, np.ndarray):
            y = torch.from_numpy(np.asarray(y, dtype=np.float32))
        if y.dim() == 1:
            y = y.unsqueeze(0)
        dev = self._ctx.device
        y = y.to(dev, torch.float32).contiguous()
        B, n = y.shape
        pad, ctr = (self.n_fft - self.hop_size) // 2, int(bool(center))
        lib = self._ctx.lib
        T = lib.vb_melnet_frames(C.byref(self._cfg), n, ctr)
        if T < 1 or pad >= n or (ctr and self.n_fft // 2 >= n + 2 * pad):
            raise ValueError(f"MelNet: {n} samples are too few for reflect padding {pad} / one {self.n_fft}-sample frame")
        need = lib.vb_melnet_workspace_bytes(C.byref(self._cfg), B, n, ctr)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        nb, off = self.n_fft // 2 + 1, im_offset(self.n_fft)
        mel = None if complex else torch.empty(B, self.num_mels, T, dtype=torch.float32, device=dev)
        spec = torch.empty(B, T, 2 * off, dtype=torch.float32, device=dev) if complex else None
        L.check(lib.vb_melnet_forward(self._ctx.handle, L.ptr(y), B, n, ctr, L.ptr(mel) if mel is not None else None,
                                      L.ptr(spec) if spec is not None else None, L.ptr(self._ws), L.stream_ptr()), "vb_melnet_forward")
        if complex:     # the reference returns view_as_real(stft).transpose(1, 2): [B, T, n_fft/2+1, 2]  (:79-81)
            return torch.stack([spec[:, :, :nb], spec[:, :, off:off + nb]], dim=-1)
        return mel
