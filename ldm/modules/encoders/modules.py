"""Target of configs/vocal2music.yaml:71 (reference: ldm/modules/encoders/modules.py:194)."""
from versband_amd.model import FrozenTextVocalEmbedder  # noqa: F401
