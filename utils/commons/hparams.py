"""reference: utils/commons/hparams.py:25 (vocoder config.yaml with base_config inheritance)."""
from versband_amd.model import set_hparams  # noqa: F401
