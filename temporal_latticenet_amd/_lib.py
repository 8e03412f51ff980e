"""ctypes binding of libtln_hip.so (the C ABI declared in include/tln.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtln_hip.so")


class TlnError(RuntimeError):
    pass


class GemmSrc(C.Structure):
    _fields_ = [
        ("d_src", C.c_void_p), ("src_rows", C.c_int64), ("ld", C.c_int64), ("cin", C.c_int), ("taps", C.c_int),
        ("d_table", C.c_void_p), ("pad_value", C.c_float), ("d_scale", C.c_void_p), ("d_shift", C.c_void_p),
        ("relu", C.c_int),
        ("d_gn_partials", C.c_void_p), ("d_gn_gamma", C.c_void_p), ("d_gn_beta", C.c_void_p), ("gn_rows", C.c_int64),
        ("gn_groups", C.c_int), ("gn_eps", C.c_float),
    ]


class GemmCall(C.Structure):
    """tln_gemm_call"""
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int), ("s0", C.POINTER(GemmSrc)), ("s1", C.POINTER(GemmSrc)), ("d_w", C.c_void_p),
        ("w_is_nk", C.c_int), ("d_bias", C.c_void_p), ("d_residual", C.c_void_p), ("ld_res", C.c_int64),
        ("relu", C.c_int), ("d_out", C.c_void_p), ("ld_out", C.c_int64), ("d_stats", C.c_void_p),
    ]


class Options(C.Structure):
    """tln_options: kernel-selection options, stored in a handle or passed with a call (never process-wide)"""
    _fields_ = [("k1_legacy", C.c_int), ("k1_bucket_rows", C.c_int), ("pool_mode", C.c_int), ("gemm_direct", C.c_int),
                ("gemm_pair_off", C.c_int), ("gemm_tn", C.c_int), ("gemm_groups", C.c_int), ("gemm_splits", C.c_int),
                ("gemm_wm", C.c_int), ("v2_off", C.c_int), ("v2_min_m", C.c_int64), ("gemm_stamps", C.c_void_p),
                ("group_off_mask", C.c_int)]


class GnDesc(C.Structure):
    _fields_ = [
        ("d_partials", C.c_void_p), ("d_x", C.c_void_p), ("V", C.c_int64), ("C", C.c_int), ("groups", C.c_int),
        ("relu", C.c_int), ("d_gamma", C.c_void_p), ("d_beta", C.c_void_p), ("eps", C.c_float),
        ("d_scale_shift", C.c_void_p), ("d_ws", C.c_void_p), ("ws_bytes", C.c_int64),
    ]


class DistributeCall(C.Structure):
    """tln_distribute_call"""
    _fields_ = [("l", C.c_void_p), ("d_positions", C.c_void_p), ("d_values", C.c_void_p), ("n", C.c_int64),
                ("val_dim", C.c_int), ("subtract_mean", C.c_int), ("d_distributed", C.c_void_p),
                ("d_indices", C.c_void_p), ("d_weights", C.c_void_p)]


class Slot(C.Structure):
    """tln_slot"""
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("kind", C.c_int), ("state", C.c_int)]


class OpSrc(C.Structure):
    """tln_op_src"""
    _fields_ = [("slot", C.c_int), ("table", C.c_int), ("level", C.c_int), ("relu", C.c_int),
                ("pad_value", C.c_float), ("gn_stats", C.c_int), ("gn_groups", C.c_int), ("gn_eps", C.c_float),
                ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p)]


class Op(C.Structure):
    """tln_op"""
    _fields_ = [("kind", C.c_int), ("cond_state", C.c_int), ("cond_has", C.c_int), ("out", C.c_int),
                ("out_col", C.c_int), ("n", C.c_int), ("stats_out", C.c_int), ("s0", OpSrc), ("s1", OpSrc),
                ("residual", C.c_int), ("relu", C.c_int), ("w_is_nk", C.c_int), ("w", C.c_void_p),
                ("bias", C.c_void_p), ("p", C.c_void_p * 8), ("i", C.c_int * 8), ("f", C.c_float * 4)]


_vp, _i64, _i, _f = C.c_void_p, C.c_int64, C.c_int, C.c_float
_PROTOS = {
    "tln_last_error": (C.c_char_p, []),
    "tln_version": (_i, []),
    "tln_lattice_create": (_i, [C.POINTER(_vp), _i, C.POINTER(C.c_double), _i64]),
    "tln_lattice_create_ex": (_i, [C.POINTER(_vp), _i, C.POINTER(C.c_double), _i64, C.c_double]),
    "tln_lattice_default_scale_constant": (C.c_double, []),
    "tln_lattice_scale_constant": (C.c_double, [_vp]),
    "tln_lattice_destroy": (_i, [_vp]),
    "tln_lattice_memory": (_i, [_vp, C.POINTER(_i64)]),
    "tln_program_memory": (_i, [_vp, C.POINTER(_i64)]),
    "tln_program_replans": (_i64, [_vp]),
    "tln_options_init": (None, [C.POINTER(Options)]),
    "tln_lattice_set_options": (_i, [_vp, C.POINTER(Options)]),
    "tln_program_set_options": (_i, [_vp, C.POINTER(Options)]),
    "tln_gather_gemm_opt": (_i, [C.POINTER(GemmCall), C.POINTER(Options), _vp]),
    "tln_gather_gemm_multi_opt": (_i, [C.POINTER(GemmCall), _i, C.POINTER(Options), _vp]),
    "tln_gn_gather_gemm_opt": (_i, [C.POINTER(GnDesc), _i64, _i, C.POINTER(GemmSrc), C.POINTER(GemmSrc), _vp, _i, _vp, _vp, _i64, _i, _vp, _i64, _vp, C.POINTER(Options), _vp]),
    "tln_gru_cell_opt": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(Options), _vp]),
    "tln_gru_cell_multi_opt": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(Options), _vp]),
    "tln_lattice_clear": (_i, [_vp, _vp]),
    "tln_lattice_clear_multi": (_i, [_vp, _i, _vp]),
    "tln_distribute_begin_multi": (_i, [_vp, _i, _vp]),
    "tln_groupnorm_partials_multi": (_i, [_vp, _i, _i, _vp]),
    "tln_aflow_multi": (_i, [_vp, _i, _i, _f, _f, _f, _i, _vp, _vp]),
    "tln_slice_deform_multi": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "tln_gru_cell_multi": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "tln_lattice_prepare_levels_begin_multi": (_i, [_vp, _i, _i, _vp, _vp]),
    "tln_lattice_prepare_levels_finish_multi": (_i, [_vp, _i, _vp]),
    "tln_gather_gemm_dw_ws_floats": (_i64, [_i64, _i, _i, _i]),
    "tln_gather_gemm_dw": (_i, [_vp, _i64, _i, _vp, _i, _vp, _i64, _i, _vp, _vp, _i64, _vp]),
    "tln_slice_blend_bwd_lv": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _i64, _vp, _vp]),
    "tln_slice_blend_bwd_w": (_i, [_vp, _i64, _i, _vp, _vp, _i64, _vp, _vp]),
    "tln_pointnet_pool_multi": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "tln_program_begin_frame_group": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "tln_lattice_nr_vertices": (_i64, [_vp]),
    "tln_lattice_capacity": (_i64, [_vp]),
    "tln_lattice_level": (_i, [_vp]),
    "tln_lattice_overflow_rows": (_i64, [_vp]),
    "tln_lattice_drop_bins": (_i, [_vp]),
    "tln_lattice_bucket_fallbacks": (_i64, [_vp]),
    "tln_lattice_keys": (_i, [_vp, _vp, _i64, _vp]),
    "tln_lattice_insert_keys": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "tln_distribute": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "tln_distribute_begin": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "tln_distribute_finish": (_i, [_vp, _vp]),
    "tln_build_csr": (_i, [_vp, _vp, _i64, _vp]),
    "tln_lattice_csr": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "tln_pointnet_pool": (_i, [_vp, _vp, _i64, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _i, _vp, _vp]),
    "tln_pointnet_pool_ex": (_i, [_vp, _vp, _i64, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _i, _vp, _vp, _vp]),
    "tln_neighbour_table": (_i, [_vp, C.POINTER(_vp), _vp]),
    "tln_coarsen": (_i, [_vp, C.POINTER(_vp), _vp]),
    "tln_lattice_prepare_levels": (_i, [_vp, _i, _vp]),
    "tln_lattice_prepare_levels_begin": (_i, [_vp, _i, C.POINTER(_i64), _vp]),
    "tln_lattice_prepare_levels_finish": (_i, [_vp, _vp]),
    "tln_lattice_coarse_level": (_vp, [_vp]),
    "tln_coarse_to_fine_table": (_i, [_vp, C.POINTER(_vp), _vp]),
    "tln_fine_to_coarse_table": (_i, [_vp, C.POINTER(_vp), _vp]),
    "tln_gather_gemm": (_i, [_i64, _i, C.POINTER(GemmSrc), C.POINTER(GemmSrc), _vp, _i, _vp, _vp, _i64, _i, _vp, _i64, _vp]),
    "tln_gather_gemm_pair": (_i, [C.POINTER(GemmCall), C.POINTER(GemmCall), _vp]),
    "tln_gather_gemm_multi": (_i, [C.POINTER(GemmCall), _i, _vp]),
    "tln_gather_gemm_ex": (_i, [_i64, _i, C.POINTER(GemmSrc), C.POINTER(GemmSrc), _vp, _i, _vp, _vp, _i64, _i, _vp, _i64, _vp, _vp]),
    "tln_gn_gather_gemm": (_i, [C.POINTER(GnDesc), _i64, _i, C.POINTER(GemmSrc), C.POINTER(GemmSrc), _vp, _i, _vp, _vp, _i64, _i, _vp, _i64, _vp, _vp]),
    "tln_groupnorm_partials": (_i, [_vp, _i64, _i, _vp, _vp]),
    "tln_groupnorm_from_partials": (_i, [_vp, _i64, _i, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "tln_im2row": (_i, [_vp, _i64, _i, _vp, _i64, _vp, _vp]),
    "tln_groupnorm_ws_bytes": (_i64, [_i64, _i]),
    "tln_groupnorm_stats": (_i, [_vp, _i64, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _i64, _vp]),
    "tln_affine_act": (_i, [_vp, _i64, _i, _vp, _vp, _i, _vp, _vp]),
    "tln_gru_cell": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tln_lstm_gates": (_i, [_vp, _i64, _i, _vp, _vp]),
    "tln_temporal_max": (_i, [_vp, _vp, _i64, _i64, _i, _f, _vp, _vp]),
    "tln_cga_gate": (_i, [_vp, _vp, _i64, _i64, _i, _f, _vp, _vp]),
    "tln_fill_empty_rows": (_i, [_vp, _i64, _i, _i, _f, _vp, _vp]),
    "tln_aflow": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _f, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "tln_slice_gather": (_i, [_vp, _i64, _i, _vp, _vp, _i64, _vp, _vp]),
    "tln_slice": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "tln_slice_deform": (_i, [_vp, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "tln_slice_deform_ls": (_i, [_vp, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "tln_program_set_aux_out": (_i, [_vp, _vp]),
    "tln_splat": (_i, [_vp, _vp, _i, _vp, _i64, _vp, _vp]),
    "tln_scatter_max": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _vp, _vp, _i64, _vp]),
    "tln_scatter_add": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _vp]),
    "tln_program_create": (_i, [C.POINTER(_vp), C.POINTER(Slot), _i, C.POINTER(Op), _i, _i, _i]),
    "tln_program_destroy": (_i, [_vp]),
    "tln_program_reset": (_i, [_vp]),
    "tln_program_begin_frame": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, C.POINTER(_i64), _vp]),
    "tln_program_begin_frame_start": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "tln_program_begin_frame_finish": (_i, [_vp, C.POINTER(_i64), _vp]),
    "tln_program_run": (_i, [_vp, _i, _vp, _i64, _i, _vp]),
    "tln_program_run_pair": (_i, [_vp, _vp, _i, _vp, _i64, _vp, _i64, _i, _vp]),
    "tln_program_run_group": (_i, [C.POINTER(C.c_void_p), _i, _i, C.POINTER(C.c_void_p), C.POINTER(_i64), _i, _vp]),
    "tln_program_capture_gemms": (_i, [_vp, _i]),
    "tln_program_timing": (_i, [_vp, _i]),
    "tln_program_timing_read": (_i, [_vp, C.POINTER(C.c_float)]),
    "tln_program_replay_gemms": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), _vp]),
    "tln_program_replay_gemms_group": (_i, [C.POINTER(_vp), _i, _i, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), _vp]),
    "tln_program_replay_executed": (_i, [C.POINTER(_vp), _i, C.POINTER(C.c_double), _vp]),
    "tln_program_frame_rows": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i64),
                                    C.POINTER(_i)]),
    "tln_program_state_info": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i)]),
    "tln_program_state_get": (_i, [_vp, _i, _vp, _vp]),
    "tln_program_state_set": (_i, [_vp, _i, _vp, _i64, _vp]),
    "tln_program_nr_ops": (_i, [_vp]),
    "tln_program_state_ops": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "tln_program_state_expect": (_i, [_vp, _i, _i64, _vp]),
    "tln_program_run_begin": (_i, [_vp, _i, _vp, _i64, _i, _vp]),
    "tln_program_run_until": (_i, [_vp, _i, _vp]),
    "tln_program_run_end": (_i, [_vp, _vp]),
    "tln_program_state_new_info": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i)]),
    "tln_program_state_get_new": (_i, [_vp, _i, _vp, _vp]),
}

_lib = None


def exported_symbols():
    return sorted(_PROTOS)


def lib():
    """Loads the shared library once; raises TlnError if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TlnError(
                "libtln_hip.so is missing (%s). Build it with `python -m temporal_latticenet_amd.build`; "
                "there is no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64; it must be in the process BEFORE our library is mapped so that both
        # share one HIP runtime (loading ours first binds it to /opt/rocm's copy, which then sees no device)
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().tln_last_error()
        raise TlnError("%s failed with code %d: %s" % (what or "tln call", rc, (msg or b"").decode()))
