"""Times the log-mel front-end (vb_melnet_forward) on B clips of 20 s: ms per call and the three kernels' share."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from versband_amd.melnet import MelNet

HP = dict(fft_size=1280, audio_num_mel_bins=80, audio_sample_rate=24000, hop_size=320, win_size=1280, fmin=0, fmax=8000)
net = MelNet(HP, device="cuda:0")
for B in (1, 8, 32):
    wav = (torch.rand(B, 1500 * 320, device="cuda") * 2 - 1) * 0.5
    for _ in range(3):
        net(wav)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        net(wav)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    flop = 2.0 * B * 1500 * 1280 * 1288
    print(f"B={B}: {ms:.3f} ms per call, {flop / ms / 1e9:.1f} TFLOP/s (fp32 MFMA conv), {B * 20 / (ms / 1e3):.0f} audio-s/s")
