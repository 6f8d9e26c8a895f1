"""`dynamic=unet_fast` target (config/dynamic/unet_fast.yaml:1) -> MI355X HIP implementation."""
from sgdm_amd.unet import UNetModel  # noqa: F401
