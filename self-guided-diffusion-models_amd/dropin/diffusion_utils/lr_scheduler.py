from sgdm_amd.util import LambdaLinearScheduler  # noqa: F401
