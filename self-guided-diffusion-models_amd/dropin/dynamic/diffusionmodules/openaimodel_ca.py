"""`dynamic=unetca_fast` target (config/dynamic/unetca_fast.yaml:1) -> MI355X HIP implementation."""
from sgdm_amd.unet import UNetModelCA as UNetModel  # noqa: F401
