"""`dynamic=unetca_fast` target (config/dynamic/unetca_fast.yaml:1) -> MI355X HIP implementation.
Other names of the reference module (openaimodel_ca.py) resolve lazily in the checkout."""
from sgdm_amd._overlay import reference_fallback
from sgdm_amd.unet import UNetModelCA as UNetModel  # noqa: F401

__getattr__ = reference_fallback(__name__, __file__)
