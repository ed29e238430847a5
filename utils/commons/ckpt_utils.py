"""reference: utils/commons/ckpt_utils.py:26 (vocoder checkpoint discovery)."""
from versband_amd.model import load_ckpt_state  # noqa: F401
