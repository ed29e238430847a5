"""Target of configs/vocal2music.yaml:46 (reference: ldm/models/autoencoder1d.py:14)."""
from versband_amd.model import AutoencoderKL  # noqa: F401
