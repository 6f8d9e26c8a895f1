from sgdm_amd.ema import LitEma  # noqa: F401
