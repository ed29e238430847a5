"""Import path of the reference's mel front-end (preprocess/NAT_mel.py:42 MelNet), served by the HIP implementation."""
from versband_amd.melnet import MelNet  # noqa: F401
