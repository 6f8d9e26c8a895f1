"""ctypes binding of libsgdm_hip.so (include/sgdm_hip.h).

The product path has NO fallback: if the library is missing or does not export a symbol the
header declares, importing/using the HIP path raises.  (Build: ``python
self-guided-diffusion-models_amd/build.py`` or ``__graft_entry__.build()``.)
"""
import ctypes as C
import os

# torch MUST be imported before the library is dlopen'ed: libsgdm_hip.so shares device pointers and
# streams with torch, so both have to run on the ONE HIP runtime (libamdhip64) torch brings into the
# process.  Loading ours first binds /opt/rocm's copy and the launches then see no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGDM_LIB_PATH") or os.path.join(_HERE, "lib", "libsgdm_hip.so")   # override: A/B builds (tools)

MODE_FLAT, MODE_CONV3 = 0, 1
RS_NONE, RS_AVGPOOL2, RS_UP2, RS_ZEROUP2 = 0, 1, 2, 3
PRO_NONE, PRO_AFFINE_NC, PRO_LN_ROW = 0, 1, 2
PREC_F32, PREC_F16X3, PREC_BF16X3 = 0, 1, 2
PREC_BY_NAME = {"f32": PREC_F32, "f16x3": PREC_F16X3, "bf16x3": PREC_BF16X3}
ABI_VERSION = 21

# sgd_igemm_args.tune (include/sgdm_hip.h: SGD_TUNE_*): per-call schedule overrides for parity tests and A/B tools
TUNE_BN128, TUNE_BN256, TUNE_FLAT2, TUNE_DEFER, TUNE_PLAIN_SCHEDULE, TUNE_LN_PACKED, TUNE_NO_SMALL = 1, 2, 4, 8, 16, 32, 64
TUNE_WGRAD_GENERIC_NARROW, TUNE_WGRAD_NO_POOLED_PLANES, TUNE_WGRAD_F32 = 256, 512, 1024
TUNE_WGRAD_NO_WS, TUNE_WGRAD_NO_PLANES, TUNE_WGRAD_NO_PIPE, TUNE_WGRAD_PLANES_ALWAYS = 2048, 4096, 8192, 16384

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class IgemmArgs(C.Structure):
    """mirror of ``struct sgd_igemm_args`` (include/sgdm_hip.h)"""
    _fields_ = [
        ("x0", vp), ("x1", vp), ("c0", i32), ("c1", i32), ("mode", i32),
        ("n", i32), ("hi", i32), ("wi", i32), ("ho", i32), ("wo", i32), ("m", i32),
        ("rows_per_n", i32), ("stride", i32), ("resample", i32), ("pro", i32), ("pro_silu", i32),
        ("pa", vp), ("pb", vp), ("pc", vp), ("w", vp), ("cin_p", i32), ("cout_p", i32),
        ("bias", vp), ("res", vp), ("res_mode", i32), ("y", vp), ("cout", i32), ("y_ld", i32),
        ("orows_in", i32), ("orows_out", i32), ("orow_off", i32), ("prec", i32),
        ("drop_p", f32), ("drop_seed", C.c_uint32), ("stats", vp), ("w_scale_inv", vp),
        ("work", vp), ("work_bytes", i64), ("grid_cap", i32), ("tune", i32),
    ]


class PackJob(C.Structure):
    """mirror of ``struct sgd_pack_job`` (include/sgdm_hip.h)"""
    _fields_ = [("src", vp), ("dst", vp), ("amax_bits", vp), ("scale_inv", vp), ("cout", i32), ("cin", i32), ("ksize", i32),
                ("transpose", i32), ("own_amax", i32), ("reserved0", i32)]


# name -> (restype, argtypes); every symbol include/sgdm_hip.h declares
SIGNATURES = {
    "sgd_abi_version": (i32, []),
    "sgd_build_id": (C.c_char_p, []),
    "sgd_igemm": (i32, [C.POINTER(IgemmArgs), vp]),
    "sgd_igemm_stats_parts": (i32, [C.POINTER(IgemmArgs)]),
    "sgd_igemm_work_bytes": (i64, []),
    "sgd_igemm_work_status_offset": (i64, []),
    "sgd_igemm_tail_layout": (i32, [i32, i32, i32, i32, C.POINTER(i32)]),
    "sgd_stats_reduce": (i32, [vp, i32, i32, i32, vp, i32, i32, vp]),
    "sgd_packed_weight_bytes": (i64, [i32, i32, i32, i32]),
    "sgd_pack_weight": (i32, [vp, vp, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), vp]),
    "sgd_weight_amax": (i32, [vp, i64, vp, vp]),
    "sgd_pack_weight_scaled": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, C.POINTER(i32), C.POINTER(i32), vp]),
    "sgd_pack_job_blocks": (i32, [i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "sgd_pack_weights_batched": (i32, [vp, i32, vp, vp, i32, vp, vp, i32, i32, vp]),
    "sgd_linear_splitk": (i32, [vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, i32, vp]),
    "sgd_linear_splitk_t": (i32, [vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp]),
    "sgd_conv3_narrow_in_parts": (i32, [i32, i32]),
    "sgd_conv3_narrow_in": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "sgd_conv3_narrow_out": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "sgd_chan_stats": (i32, [vp, i32, i32, i32, vp, i32, i32, vp]),
    "sgd_gn_coef_parts": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "sgd_gn_coef": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp]),
    "sgd_add_rows_nc": (i32, [vp, vp, i32, i32, i64, i32, vp]),
    "sgd_ln_stats": (i32, [vp, i32, i32, f32, vp, vp]),
    "sgd_ln_apply": (i32, [vp, vp, vp, vp, i32, i32, f32, vp, vp]),
    "sgd_attention": (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp, vp]),
    "sgd_attention_split": (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp, vp]),
    "sgd_attention_masked": (i32, [vp, i32, i32, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, f32, vp, i32, vp, vp]),
    "sgd_linear_attention": (i32, [vp, i32, i32, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, f32, vp, i32, vp]),
    "sgd_attention_bwd": (i32, [vp, i32, i32, vp, vp, i32, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32,
                              vp, vp, vp, vp]),
    "sgd_attention_bwd_split": (i32, [vp, i32, i32, vp, vp, i32, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32,
                                    vp, vp, vp, vp]),
    "sgd_attention_masked_bwd": (i32, [vp, i32, i32, vp, vp, i32, i32, vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32,
                                     f32, vp, vp, vp, vp]),
    "sgd_linear_attention_bwd": (i32, [vp, i32, i32, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp,
                                     vp]),
    "sgd_pack_weight_dgrad": (i32, [vp, vp, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), vp]),
    "sgd_wgrad": (i32, [C.POINTER(IgemmArgs), vp, i32, i32, vp, i32, vp, vp]),
    "sgd_wgrad_scratch_bytes": (i64, [C.POINTER(IgemmArgs), i32]),
    "sgd_wgrad_scratch": (i32, [C.POINTER(IgemmArgs), vp, i32, i32, vp, i32, vp, vp, i64, vp]),
    "sgd_colsum_fold": (i32, [vp, i32, i32, vp, i32, f32, vp]),
    "sgd_wgrad_reduce": (i32, [vp, i32, i32, i32, i32, vp, i32, f32, vp]),
    "sgd_wgrad_reduce_bias": (i32, [vp, i32, i32, i32, i32, vp, i32, f32, vp, i32, vp, vp]),
    "sgd_wgrad_bias_rows": (i32, [C.POINTER(IgemmArgs), i32, i32, i32, i64]),
    "sgd_colsum_pair": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, f32, vp]),
    "sgd_colsum": (i32, [vp, i32, i32, i32, vp, i32, f32, vp, i32, vp]),
    "sgd_gn_bwd_reduce": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, f32, C.c_uint32, vp, vp]),
    "sgd_gn_bwd_coef": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "sgd_gn_bwd_coef_fold": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, i32, f32, vp]),
    "sgd_gn_bwd_rows_chunks": (i32, [i32, i32, i32, i32]),
    "sgd_gn_bwd_reduce_rows": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, f32, C.c_uint32, i32, vp, vp]),
    "sgd_gn_bwd_apply": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, f32, C.c_uint32, vp, vp, vp, vp, i32, i32,
                             vp, i32, i32, i32, vp]),
    "sgd_silu_bwd": (i32, [vp, vp, i64, vp, vp]),
    "sgd_ln_bwd": (i32, [vp, vp, vp, i32, i32, f32, vp, i32, vp, vp, vp]),
    "sgd_resample_bwd": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "sgd_q_sample": (i32, [vp, vp, vp, vp, vp, i32, i64, vp, vp]),
    "sgd_mse_loss": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "sgd_adamw_ema_step": (i32, [vp, vp, i32, i32, f32, f32, f32, f32, f32, f32, f32, f32, f32, vp, vp]),
    "sgd_exchange_bind": (i32, [C.c_char_p]),
    "sgd_allreduce_bucket": (i32, [vp, vp, i64, vp]),
    "sgd_timestep_embedding": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "sgd_cond_select": (i32, [vp, i32, vp, vp, i32, i32, i32, vp, vp]),
    "sgd_pack_input": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sgd_pack_input_compact": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sgd_labelmap_nhot": (i32, [vp, i32, i32, i32, vp, vp]),
    "sgd_linear_gather": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "sgd_linear_sparse_rows": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "sgd_nhwc_to_nchw": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "sgd_fill_null_kv": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "sgd_ddpm_step": (i32, [vp, vp, vp, i32, f32, C.POINTER(f32), i32, i32, i32, i32, vp, vp, vp]),
    "sgd_ddim_step": (i32, [vp, vp, vp, i32, f32, C.POINTER(f32), f32, i32, i32, i32, i32, vp, vp, vp]),
    "sgd_ddpm_step_dev": (i32, [vp, vp, vp, i32, f32, vp, i32, i32, i32, i32, vp, vp, vp]),
    "sgd_ddim_step_dev": (i32, [vp, vp, vp, i32, f32, vp, f32, i32, i32, i32, i32, vp, vp, vp]),
    "sgd_x0_quantile": (i32, [i32, vp, vp, i32, f32, C.POINTER(f32), i32, i32, i32, i32, i32, f32, vp, vp]),
    "sgd_ddpm_step_dyn": (i32, [vp, vp, vp, i32, f32, C.POINTER(f32), vp, i32, i32, i32, vp, vp, vp]),
    "sgd_ddim_step_dyn": (i32, [vp, vp, vp, i32, f32, C.POINTER(f32), f32, vp, i32, i32, i32, vp, vp, vp]),
    "sgd_token_pool": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "sgd_geglu": (i32, [vp, i64, i32, vp, vp]),
    "sgd_to_uint8": (i32, [vp, i64, vp, vp]),
    "sgd_cfg_combine": (i32, [vp, i32, f32, i32, i32, i32, vp, vp]),
}

# include/sgdm_hip_tools.h: the diagnostics library (libsgdm_hip_tools.so) -- bench.py's device calibration, the contention
# tests, tools/.  NOT part of the product library and never loaded by sgdm_amd/ itself.
TOOLS_LIB_PATH = os.path.join(_HERE, "lib", "libsgdm_hip_tools.so")
TOOLS_SIGNATURES = {
    "sgd_debug_occupy": (i32, [i32, f32, vp]),
    "sgd_debug_mfma_probe": (i32, [i32, i64, C.c_uint32, i32, vp, vp]),
    "sgd_debug_mfma_probe_flops": (i64, [i32, i64, i32]),
    "sgd_debug_copy_probe": (i32, [vp, vp, i64, i32, i32, vp]),
    "sgd_debug_mfma_lds_probe": (i32, [i32, i64, C.c_uint32, i32, i32, vp, vp]),
    "sgd_debug_mfma_stream_probe": (i32, [i32, i64, C.c_uint32, i32, vp, vp, i64, vp, vp]),
}

_lib = None
_tools = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the library (once) and bind every declared symbol; raises HipLibraryError loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run `python self-guided-diffusion-models_amd/build.py`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.sgd_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library {lib.sgd_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def load_tools():
    """the diagnostics library (tests / bench / tools only)"""
    global _tools
    if _tools is not None:
        return _tools
    if not os.path.exists(TOOLS_LIB_PATH):
        raise HipLibraryError(f"{TOOLS_LIB_PATH} not found (run `python self-guided-diffusion-models_amd/build.py`)")
    lib = C.CDLL(TOOLS_LIB_PATH)
    for name, (res, args) in TOOLS_SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{TOOLS_LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _tools = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc} "
                           f"({'invalid argument' if rc == 1 else 'launch failure'})")
