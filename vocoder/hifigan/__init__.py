"""reference: vocoder/hifigan/__init__.py (scripts/test_final.py:21 imports HifiGAN from here)."""
from versband_amd.model import HifiGAN  # noqa: F401
