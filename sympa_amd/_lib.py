"""ctypes binding of the C-ABI in include/sympa_hip.h.

The product path has NO fallback: if libsympa_hip.so is missing or does not export a declared
symbol, importing/using the ops raises.  (Build it with `python -c "import __graft_entry__ as g; g.build()"`;
`make -C sympa_amd/csrc` runs the same function -- there is ONE build path, the one that scans every translation
unit's ISA for the DPP hazard and rebuilds the units that show it.)"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SYMPA_HIP_LIB: another BUILD of the same library (tools/build_variant.sh: A/B timing, compiler-flag experiments); never a
# different implementation -- the symbol check below applies to it as well
LIB_PATH = os.environ.get("SYMPA_HIP_LIB") or os.path.join(_HERE, "csrc", "libsympa_hip.so")

# every symbol include/sympa_hip.h declares (tests check the header against this list)
SYMBOLS = (
    "sympa_version",
    "sympa_last_error",
    "sympa_max_dims",
    "sympa_set_instance_fallback",
    "sympa_get_instance_fallback",
    "sympa_siegel_dist_fwd",
    "sympa_model_forward",
    "sympa_model_forward_batches",
    "sympa_table_pack_bytes",
    "sympa_table_pack",
    "sympa_table_digest",
    "sympa_clock_stamp",
    "sympa_table_pack_refresh",
    "sympa_model_forward_packed",
    "sympa_model_forward_batches_packed",
    "sympa_all_pairs_dist",
    "sympa_all_pairs_workspace_bytes",
    "sympa_all_pairs_dist_packed",
    "sympa_siegel_backward_workspace_bytes",
    "sympa_siegel_dist_bwd",
    "sympa_model_backward",
    "sympa_model_loss_backward",
    "sympa_model_loss_backward_rows",
    "sympa_scatter_add_rows",
    "sympa_egrad2rgrad",
    "sympa_tangent_sqnorm",
    "sympa_projx",
    "sympa_rsgd_step",
    "sympa_radam_step",
    "sympa_radam_step_fused",
    "sympa_sqnorm_accum",
    "sympa_sgd_step_clipped",
    "sympa_rsgd_step_clipped",
    "sympa_model_train_backward",
    "sympa_segment_sum_rows",
    "sympa_segment_sum_partials",
    "sympa_rsgd_step_fused_workspace_bytes",
    "sympa_rsgd_step_fused",
    "sympa_spd_dist_fwd",
    "sympa_spd_model_forward",
    "sympa_spd_table_pack_bytes",
    "sympa_spd_table_pack",
    "sympa_spd_table_pack_refresh",
    "sympa_spd_model_forward_packed",
    "sympa_scatter_add_flat_rows",
    "sympa_spd_backward_rows",
    "sympa_spd_backward_workspace_bytes",
    "sympa_spd_loss_backward",
    "sympa_spd_egrad2rgrad",
    "sympa_spd_projx",
    "sympa_spd_rsgd_step",
)

_c_double_p = ctypes.c_void_p
_c_i64_p = ctypes.c_void_p
_c_i32_p = ctypes.c_void_p

_lib = None
_fast = None          # sympa_amd/_fast.<abi>.so (csrc/torch_binding.cpp) once bound; False when absent or switched off


class SympaHipError(RuntimeError):
    pass


def fast():
    """The thin torch binding of sympa_model_forward (csrc/torch_binding.cpp: tensors in, one C call), bound to the SAME
    loaded libsympa_hip.so as the ctypes prototypes -- or None when it is not built (`__graft_entry__.build()` builds it) or
    SYMPA_NO_FAST_BINDING is set: callers then go through ctypes, which calls the same C-ABI entry of the same library."""
    global _fast
    if _fast is None:
        _fast = False
        if not os.environ.get("SYMPA_NO_FAST_BINDING"):
            try:
                import torch  # noqa: F401  (the extension links against libtorch)
                from sympa_amd import _fast as mod
                lib = load()
                mod.bind(ctypes.cast(lib.sympa_model_forward, ctypes.c_void_p).value,
                         ctypes.cast(lib.sympa_last_error, ctypes.c_void_p).value)
                _fast = mod
            except ImportError:
                _fast = False
    return _fast or None


def load():
    """Loads libsympa_hip.so once and sets the prototypes.  Raises SympaHipError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SympaHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; "
            "run __graft_entry__.build() (hipcc --offload-arch=gfx950)."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise SympaHipError(f"{LIB_PATH} does not export {s}")
    lib.sympa_version.restype = ctypes.c_char_p
    lib.sympa_last_error.restype = ctypes.c_char_p
    lib.sympa_max_dims.restype = ctypes.c_int
    lib.sympa_set_instance_fallback.restype = ctypes.c_int
    lib.sympa_set_instance_fallback.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.sympa_get_instance_fallback.restype = ctypes.c_int
    lib.sympa_get_instance_fallback.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.sympa_siegel_dist_fwd.restype = ctypes.c_int
    lib.sympa_siegel_dist_fwd.argtypes = [
        _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        _c_double_p, ctypes.c_double, _c_double_p, _c_double_p, _c_i32_p, ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_model_forward.restype = ctypes.c_int
    lib.sympa_model_forward.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, _c_i64_p, ctypes.c_int64, _c_i64_p, ctypes.c_int64,
        ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_double, _c_double_p,
        ctypes.c_double, _c_double_p, _c_i32_p, ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_model_forward_batches.restype = ctypes.c_int
    lib.sympa_model_forward_batches.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double, ctypes.c_void_p,
        _c_i32_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
    ]
    lib.sympa_all_pairs_dist.restype = ctypes.c_int
    lib.sympa_all_pairs_dist.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
        _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double, _c_double_p, _c_i32_p, ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_all_pairs_workspace_bytes.restype = ctypes.c_int64
    lib.sympa_all_pairs_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    lib.sympa_all_pairs_dist_packed.restype = ctypes.c_int
    lib.sympa_all_pairs_dist_packed.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
        _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_void_p, ctypes.c_int64,
        _c_i32_p, ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_siegel_dist_bwd.restype = ctypes.c_int
    lib.sympa_siegel_dist_bwd.argtypes = [
        _c_double_p, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        _c_double_p, ctypes.c_double, _c_double_p, _c_double_p, _c_double_p, _c_i32_p, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_siegel_backward_workspace_bytes.restype = ctypes.c_int64
    lib.sympa_siegel_backward_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    lib.sympa_model_backward.restype = ctypes.c_int
    lib.sympa_model_backward.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, _c_i64_p, ctypes.c_int64, _c_i64_p, ctypes.c_int64,
        ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double,
        _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_i32_p, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_model_loss_backward.restype = ctypes.c_int
    lib.sympa_model_loss_backward.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, _c_i64_p, ctypes.c_int64, _c_i64_p, ctypes.c_int64, _c_double_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double,
        ctypes.c_double, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_i32_p, ctypes.c_void_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_model_loss_backward_rows.restype = ctypes.c_int
    lib.sympa_model_loss_backward_rows.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, _c_i64_p, ctypes.c_int64, _c_i64_p, ctypes.c_int64, _c_double_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_double, _c_double_p, ctypes.c_double,
        ctypes.c_double, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_i32_p,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_scatter_add_rows.restype = ctypes.c_int
    lib.sympa_scatter_add_rows.argtypes = [_c_double_p, _c_i64_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_int64, ctypes.c_double, _c_double_p, _c_i32_p, ctypes.c_void_p]
    lib.sympa_egrad2rgrad.restype = ctypes.c_int
    lib.sympa_egrad2rgrad.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_double_p,
                                      ctypes.c_void_p]
    lib.sympa_tangent_sqnorm.restype = ctypes.c_int
    lib.sympa_tangent_sqnorm.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_double_p,
                                         _c_i32_p, ctypes.c_void_p]
    lib.sympa_projx.restype = ctypes.c_int
    lib.sympa_projx.argtypes = [_c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double, _c_double_p,
                                _c_i32_p, _c_i32_p, _c_i32_p, ctypes.c_void_p]
    lib.sympa_model_train_backward.restype = ctypes.c_int
    lib.sympa_model_train_backward.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, _c_i64_p, ctypes.c_int64, _c_i64_p, ctypes.c_int64, _c_double_p,
        ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _c_double_p, ctypes.c_double, _c_double_p,
        ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p,
        _c_i32_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_segment_sum_rows.restype = ctypes.c_int
    lib.sympa_segment_sum_rows.argtypes = [
        _c_double_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_double, ctypes.c_int, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_double_p,
        _c_double_p, _c_double_p, _c_double_p, ctypes.c_void_p,
    ]
    lib.sympa_segment_sum_partials.restype = ctypes.c_int64
    lib.sympa_segment_sum_partials.argtypes = [ctypes.c_int64, ctypes.c_int]
    lib.sympa_rsgd_step_fused_workspace_bytes.restype = ctypes.c_int64
    lib.sympa_rsgd_step_fused_workspace_bytes.argtypes = [ctypes.c_int64]
    lib.sympa_radam_step_fused.restype = ctypes.c_int
    lib.sympa_radam_step_fused.argtypes = [
        _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
        ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, _c_double_p, ctypes.c_int,
        ctypes.c_void_p, _c_i32_p, _c_i32_p, ctypes.c_void_p,
    ]
    lib.sympa_rsgd_step_fused.restype = ctypes.c_int
    lib.sympa_rsgd_step_fused.argtypes = [
        _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
        ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, _c_double_p, ctypes.c_int, ctypes.c_void_p, _c_i32_p,
        _c_i32_p, ctypes.c_void_p,
    ]
    lib.sympa_radam_step.restype = ctypes.c_int
    lib.sympa_radam_step.argtypes = [_c_double_p, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     _c_double_p, ctypes.c_double, _c_i32_p, _c_i32_p, ctypes.c_void_p]
    lib.sympa_rsgd_step.restype = ctypes.c_int
    lib.sympa_rsgd_step.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_double, _c_i32_p, _c_i32_p, _c_i32_p, ctypes.c_void_p]
    lib.sympa_sqnorm_accum.restype = ctypes.c_int
    lib.sympa_sqnorm_accum.argtypes = [_c_double_p, ctypes.c_int64, _c_double_p, ctypes.c_void_p]
    lib.sympa_sgd_step_clipped.restype = ctypes.c_int
    lib.sympa_sgd_step_clipped.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                           _c_double_p, ctypes.c_double, ctypes.c_void_p]
    lib.sympa_rsgd_step_clipped.restype = ctypes.c_int
    lib.sympa_rsgd_step_clipped.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_double, ctypes.c_double, _c_double_p,
                                            ctypes.c_double, _c_i32_p, _c_i32_p, _c_i32_p, ctypes.c_void_p]
    lib.sympa_spd_dist_fwd.restype = ctypes.c_int
    lib.sympa_spd_dist_fwd.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, _c_double_p, _c_i32_p,
                                       ctypes.c_int, ctypes.c_void_p]
    lib.sympa_spd_model_forward.restype = ctypes.c_int
    lib.sympa_spd_model_forward.argtypes = [_c_double_p, ctypes.c_int64, ctypes.c_int, _c_i64_p, ctypes.c_int64, _c_i64_p,
                                            ctypes.c_int64, ctypes.c_int64, _c_double_p, ctypes.c_double, _c_double_p,
                                            _c_i32_p, ctypes.c_int, ctypes.c_void_p]
    lib.sympa_scatter_add_flat_rows.restype = ctypes.c_int
    lib.sympa_scatter_add_flat_rows.argtypes = [_c_double_p, _c_i64_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_int64, ctypes.c_double, _c_double_p, _c_i32_p, ctypes.c_void_p]
    lib.sympa_spd_backward_rows.restype = ctypes.c_int
    lib.sympa_spd_backward_rows.argtypes = [
        _c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, _c_i64_p, ctypes.c_int64, _c_i64_p, ctypes.c_int64,
        ctypes.c_int64, _c_double_p, ctypes.c_double, _c_double_p, _c_double_p, ctypes.c_double, _c_double_p,
        _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_i32_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
        ctypes.c_void_p,
    ]
    lib.sympa_spd_backward_workspace_bytes.restype = ctypes.c_int64
    lib.sympa_spd_backward_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
    lib.sympa_spd_loss_backward.restype = ctypes.c_int
    lib.sympa_spd_loss_backward.argtypes = [
        _c_double_p, ctypes.c_int64, ctypes.c_int, _c_i64_p, ctypes.c_int64, _c_i64_p, ctypes.c_int64, ctypes.c_int64,
        _c_double_p, ctypes.c_double, _c_double_p, _c_double_p, ctypes.c_double, _c_double_p, _c_double_p, _c_double_p,
        _c_double_p, _c_i32_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
    ]
    lib.sympa_spd_egrad2rgrad.restype = ctypes.c_int
    lib.sympa_spd_egrad2rgrad.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, _c_double_p, ctypes.c_void_p]
    lib.sympa_spd_projx.restype = ctypes.c_int
    lib.sympa_spd_projx.argtypes = [_c_double_p, ctypes.c_int64, ctypes.c_int, _c_double_p, _c_i32_p, _c_i32_p, ctypes.c_void_p]
    lib.sympa_spd_rsgd_step.restype = ctypes.c_int
    lib.sympa_spd_rsgd_step.argtypes = [_c_double_p, _c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_double, _c_double_p, ctypes.c_double, _c_i32_p, ctypes.c_void_p]
    C = ctypes
    lib.sympa_table_pack_bytes.restype = C.c_int64
    lib.sympa_table_pack_bytes.argtypes = [C.c_int64, C.c_int, C.c_int]
    lib.sympa_table_pack.restype = C.c_int
    lib.sympa_table_pack.argtypes = [_c_double_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, _c_i32_p, C.c_void_p]
    lib.sympa_clock_stamp.restype = C.c_int
    lib.sympa_clock_stamp.argtypes = [C.c_void_p, C.c_void_p]
    lib.sympa_table_digest.restype = C.c_int
    lib.sympa_table_digest.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    lib.sympa_table_pack_refresh.restype = C.c_int
    lib.sympa_table_pack_refresh.argtypes = [_c_double_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                             _c_i32_p, C.c_void_p]
    lib.sympa_spd_table_pack_refresh.restype = C.c_int
    lib.sympa_spd_table_pack_refresh.argtypes = [_c_double_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, _c_i32_p,
                                                 C.c_void_p]
    lib.sympa_model_forward_packed.restype = C.c_int
    lib.sympa_model_forward_packed.argtypes = [
        C.c_void_p, C.c_int64, C.c_int64, C.c_int, _c_i64_p, C.c_int64, _c_i64_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
        _c_double_p, C.c_double, _c_double_p, C.c_double, _c_double_p, _c_i32_p, C.c_int, C.c_void_p]
    lib.sympa_model_forward_batches_packed.restype = C.c_int
    lib.sympa_model_forward_batches_packed.argtypes = [
        C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int,
        _c_double_p, C.c_double, _c_double_p, C.c_double, C.c_void_p, _c_i32_p, C.c_int, C.c_void_p]
    lib.sympa_spd_table_pack_bytes.restype = C.c_int64
    lib.sympa_spd_table_pack_bytes.argtypes = [C.c_int64, C.c_int]
    lib.sympa_spd_table_pack.restype = C.c_int
    lib.sympa_spd_table_pack.argtypes = [_c_double_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, _c_i32_p, C.c_void_p]
    lib.sympa_spd_model_forward_packed.restype = C.c_int
    lib.sympa_spd_model_forward_packed.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, _c_i64_p, C.c_int64, _c_i64_p,
                                                   C.c_int64, C.c_int64, _c_double_p, C.c_double, _c_double_p, _c_i32_p, C.c_int,
                                                   C.c_void_p]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().sympa_last_error().decode()
        raise SympaHipError(f"sympa_hip call failed (code {rc}): {msg}")
