"""ldm.util shim (reference: ldm/util.py:110-125) -> versband_amd.model factory."""
from versband_amd.model import get_obj_from_str, instantiate_from_config  # noqa: F401
