"""Target of configs/vocal2music.yaml:3 (reference: ldm/models/diffusion/cfm1_audio.py:31)."""
from versband_amd.model import CFM  # noqa: F401
