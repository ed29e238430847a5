"""Target of configs/vocal2music.yaml:34 (reference: ldm/modules/diffusionmodules/vocal2music_moe.py:477)."""
from versband_amd.model import TxtFlagLargeImprovedDiTV2  # noqa: F401
