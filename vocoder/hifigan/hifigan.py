"""Target of configs/vocal2music.yaml:89 (reference: vocoder/hifigan/hifigan.py:7)."""
from versband_amd.model import HifiGAN  # noqa: F401
