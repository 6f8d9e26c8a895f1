"""`model.target: diffusion.ddpm.LatentDiffusion` (config/model/ddpm.yaml:1) -> fused sampler steps / HIP training step."""
from sgdm_amd._overlay import reference_fallback
from sgdm_amd.diffusion import LatentDiffusion  # noqa: F401

__getattr__ = reference_fallback(__name__, __file__)
