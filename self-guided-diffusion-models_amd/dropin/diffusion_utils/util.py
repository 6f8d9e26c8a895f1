from sgdm_amd.util import instantiate_from_config  # noqa: F401
