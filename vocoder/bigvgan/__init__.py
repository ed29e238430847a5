"""Shim package: the reference's vocoder/bigvgan/__init__.py re-exports VocoderBigVGAN."""
from vocoder.bigvgan.models import VocoderBigVGAN  # noqa: F401
