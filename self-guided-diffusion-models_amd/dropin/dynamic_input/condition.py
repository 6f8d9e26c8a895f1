"""`dynamic_input.condition` (lightning_module.py:21-24, dynamic_input/condition.py:5-157) -> table-driven plugin."""
from sgdm_amd._overlay import reference_fallback
from sgdm_amd.plugin import (prepare_condition_kwargs, prepare_denoise_fn_kwargs_4sampling,  # noqa: F401
                             prepare_denoise_fn_kwargs_4sharestep, randomsample_cond)

__getattr__ = reference_fallback(__name__, __file__)
