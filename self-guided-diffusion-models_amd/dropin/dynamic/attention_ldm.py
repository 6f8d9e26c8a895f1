"""`dynamic.attention_ldm`: CrossAttention / LinearCrossAttention on the HIP kernels; every other name of the
reference module (`log` -- the only one the reference itself imports --, Attention, PerceiverResampler, ...) resolves lazily
in the checkout."""
from sgdm_amd._overlay import reference_fallback
from sgdm_amd.attention_ldm import CrossAttention, LinearCrossAttention  # noqa: F401

__getattr__ = reference_fallback(__name__, __file__)
