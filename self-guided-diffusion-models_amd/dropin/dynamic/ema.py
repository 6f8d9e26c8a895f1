"""`dynamic.ema.LitEma` (lightning_module.py:35,64-66) -> version-bumping copy_to / restore (see sgdm_amd/ema.py)."""
from sgdm_amd._overlay import reference_fallback
from sgdm_amd.ema import LitEma  # noqa: F401

__getattr__ = reference_fallback(__name__, __file__)
