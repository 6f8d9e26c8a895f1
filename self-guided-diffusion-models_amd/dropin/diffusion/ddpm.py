"""`model.target: diffusion.ddpm.LatentDiffusion` (config/model/ddpm.yaml:1)."""
from sgdm_amd.diffusion import LatentDiffusion  # noqa: F401
