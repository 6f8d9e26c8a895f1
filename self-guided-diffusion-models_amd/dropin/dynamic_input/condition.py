from sgdm_amd.plugin import (prepare_condition_kwargs, prepare_denoise_fn_kwargs_4sampling,  # noqa: F401
                             prepare_denoise_fn_kwargs_4sharestep, randomsample_cond)
