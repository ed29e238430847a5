"""Target `vocoder.bigvgan.models.VocoderBigVGAN` (reference: vocoder/bigvgan/models.py:393-414, configs/ae_accomp.yaml:51)."""
from versband_amd.model import VocoderBigVGAN  # noqa: F401
