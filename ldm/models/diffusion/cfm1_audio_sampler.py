"""reference: ldm/models/diffusion/cfm1_audio_sampler.py:26 (imported by scripts/test_final.py:23)."""
from versband_amd.model import CFMSampler  # noqa: F401
