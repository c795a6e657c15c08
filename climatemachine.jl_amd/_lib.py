"""ctypes binding of ``libcmdg.so`` (the C ABI of ``include/cmdg.h``).

The product path has no CPU fallback: if the HIP library is missing or does not
export a declared symbol this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMDG_LIB", os.path.join(_HERE, "libcmdg.so"))   # CMDG_LIB: tuning builds

CMDG_K = dict(GRADIENTS=0, DIVGRAD=1, GRADLAP=2, TENDENCY=3, PACK=4, UNPACK=5, UPDATE_AUX=6,
              FILTER=7, STACK_INTEGRAL=8, TRANSPORT=9, HALO_EXPOSED=10, GRADIENTS_EXT=11,
              DIVGRAD_EXT=12, GRADLAP_EXT=13, TENDENCY_EXT=14)
STACK_MAXOUT = 8
OPT_KEEP_GRADFLUX = 1
OPT_STACK_HEIGHT = 2
OPT_REFERENCE_HALO = 3
OPT_HALO_PIPELINE = 4
OPT_STEP_GRAPH = 5
OPT_STREAM_PRIORITY = 6
OPT_TENDENCY_PAIRS = 7
OPT_ASYNC_RUN = 8
OPT_TENDENCY_FOUR_WAVES = 9
CMDG_Q = dict(GRADFLUX_LIVE=1, LAW_NEEDS_GRADFLUX=2, NDERIVED=3, NUPDATED_AUX=4, FUSED_UPDATE_AUX=5,
              DIRECT_SEND=6, DIRECT_RECV=7, TENDENCY_ELEMS_PER_GROUP=8, HALO_PIPELINE=9, HOST_POST_NS=10, HOST_POST_COUNT=11, GRAPH_STEPS=12, TENDENCY_PAIRS=13,
              STATE_READ=16, AUX_READ=20)


class CmdgStackIntegralDesc(C.Structure):
    """``cmdg_stack_integral_desc`` of include/cmdg.h."""
    _fields_ = [
        ("nout", C.c_int32),
        ("src_is_state", C.c_int32 * STACK_MAXOUT), ("src_col", C.c_int32 * STACK_MAXOUT),
        ("scale", C.c_double * STACK_MAXOUT), ("dst_col", C.c_int32 * STACK_MAXOUT),
        ("rsrc_col", C.c_int32 * STACK_MAXOUT), ("rdst_col", C.c_int32 * STACK_MAXOUT),
    ]


class CmdgDesc(C.Structure):
    """``cmdg_desc`` of include/cmdg.h."""
    _fields_ = [
        ("dim", C.c_int32), ("N", C.c_int32 * 3),
        ("nreal", C.c_int64), ("nghost", C.c_int64),
        ("nvgeo", C.c_int32), ("physics_id", C.c_int32),
        ("iparam", C.c_int32 * 16), ("dparam", C.c_double * 64),
        ("nf_first", C.c_int32), ("direction", C.c_int32),
        ("diffusion_direction", C.c_int32), ("stacked", C.c_int32),
        ("vgeo", C.c_void_p), ("sgeo", C.c_void_p),
        ("vmapM", C.c_void_p), ("vmapP", C.c_void_p), ("elemtobndy", C.c_void_p),
        ("interiorelems", C.c_void_p), ("ninterior", C.c_int64),
        ("exteriorelems", C.c_void_p), ("nexterior", C.c_int64),
        ("activedofs", C.c_void_p), ("D", C.c_void_p),
        ("vmapsend", C.c_void_p), ("nvmapsend", C.c_int64),
        ("vmaprecv", C.c_void_p), ("nvmaprecv", C.c_int64),
        ("nnabr", C.c_int32), ("nabrtorank", C.c_void_p),
        ("nabrtovmapsend", C.c_void_p), ("nabrtovmaprecv", C.c_void_p),
        ("state_auxiliary", C.c_void_p), ("state_gradient_flux", C.c_void_p),
        ("Qhypervisc_grad", C.c_void_p), ("Qhypervisc_div", C.c_void_p),
        ("Dv", C.c_void_p),
    ]


MAX_HOOK_OPS = 4


class CmdgRhsHooks(C.Structure):
    """``cmdg_rhs_hooks`` of include/cmdg.h."""
    _fields_ = [
        ("npre", C.c_int32), ("pre_filter", C.c_void_p * MAX_HOOK_OPS),
        ("ncopy", C.c_int32),
        ("copy_gf_col", C.c_int32 * MAX_HOOK_OPS), ("copy_aux_col", C.c_int32 * MAX_HOOK_OPS),
        ("copy_scale", C.c_double * MAX_HOOK_OPS),
        ("has_integral", C.c_int32), ("has_reverse_integral", C.c_int32),
        ("integral", CmdgStackIntegralDesc), ("reverse_integral", CmdgStackIntegralDesc),
        ("nsurf", C.c_int32),
        ("surf_src_col", C.c_int32 * MAX_HOOK_OPS), ("surf_dst_col", C.c_int32 * MAX_HOOK_OPS),
        ("nvertelem", C.c_int32), ("Imat", C.c_void_p),
        ("has_flow_deviation", C.c_int32), ("flow_u_col", C.c_int32), ("flow_ud_col", C.c_int32),
        ("flow_H", C.c_double),
        ("ops_before_gradients", C.c_int32), ("pre_rhs_handle", C.c_void_p),
        ("pre_rhs_src_col", C.c_int32), ("pre_rhs_dst_aux_col", C.c_int32),
    ]


class CmdgOcean01Desc(C.Structure):
    """``cmdg_ocean01_desc`` of include/cmdg.h."""
    _fields_ = [("nvertelem", C.c_int32), ("H", C.c_double), ("Imat", C.c_void_p),
                ("add_fast_substeps", C.c_int32), ("nstages_fast", C.c_int32),
                ("rka_fast", C.c_void_p), ("rkb_fast", C.c_void_p), ("rkc_fast", C.c_void_p)]


class CmdgOceanCouplingDesc(C.Structure):
    """``cmdg_ocean_coupling_desc`` of include/cmdg.h."""
    _fields_ = [
        ("nvertelem", C.c_int32), ("H", C.c_double), ("Imat", C.c_void_p),
        ("slow_u_col", C.c_int32), ("slow_eta_col", C.c_int32), ("slow_dGu_col", C.c_int32),
        ("fast_eta_col", C.c_int32), ("fast_U_col", C.c_int32),
        ("fast_GU_col", C.c_int32), ("fast_du_col", C.c_int32),
    ]


# every symbol include/cmdg.h declares: (name, restype, argtypes)
_vp, _i32, _i64, _d = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SYMBOLS = [
    ("cmdg_version", C.c_char_p, []),
    ("cmdg_status_string", C.c_char_p, [C.c_int]),
    ("cmdg_physics_counts", C.c_int, [_i32, _vp, _vp]),
    ("cmdg_create", C.c_int, [C.POINTER(CmdgDesc), C.POINTER(_vp)]),
    ("cmdg_destroy", C.c_int, [_vp]),
    ("cmdg_last_error", C.c_char_p, [_vp]),
    ("cmdg_rhs", C.c_int, [_vp, _vp, _vp, _d, _d, _d]),
    ("cmdg_rhs_async", C.c_int, [_vp, _vp, _vp, _d, _d, _d]),
    ("cmdg_lsrk_step", C.c_int, [_vp, _vp, _vp, _d, _d, _i32, _vp, _vp, _vp]),
    ("cmdg_lsrk_run", C.c_int, [_vp, _vp, _vp, _d, _d, _i64, _i32, _vp, _vp, _vp]),
    ("cmdg_synchronize", C.c_int, [_vp]),
    ("cmdg_set_option", C.c_int, [_vp, _i32, _i32]),
    ("cmdg_query", C.c_int, [_vp, _i32, _vp]),
    ("cmdg_export_hypervisc_grad", C.c_int, [_vp, _vp]),
    ("cmdg_export_gradient_flux", C.c_int, [_vp, _vp]),
    ("cmdg_halo_begin", C.c_int, [_vp, _vp, _i32]),
    ("cmdg_halo_end", C.c_int, [_vp, _vp, _i32]),
    ("cmdg_fillsendbuf", C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32]),
    ("cmdg_transferrecvbuf", C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32]),
    ("cmdg_comm_unique_id", C.c_int, [_vp]),
    ("cmdg_comm_init_rccl", C.c_int, [_vp, _vp, _i32, _i32]),
    ("cmdg_comm_selftest", C.c_int, [_vp, _i64]),
    ("cmdg_comm_connect_local", C.c_int, [_vp, _i32]),
    ("cmdg_group_rhs", C.c_int, [_vp, _i32, _vp, _vp, _d, _d, _d]),
    ("cmdg_group_halo", C.c_int, [_vp, _i32, _vp, _i32]),
    ("cmdg_group_lsrk_run", C.c_int, [_vp, _i32, _vp, _vp, _d, _d, _i64, _i32, _vp, _vp, _vp]),
    ("cmdg_norm2_local", C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    ("cmdg_distance2_local", C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    ("cmdg_courant", C.c_int, [_vp, _i32, _vp, _d, _d, _i32, _vp]),
    ("cmdg_min_node_distance", C.c_int, [_vp, _i32, _vp]),
    ("cmdg_indefinite_stack_integral", C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    ("cmdg_reverse_indefinite_stack_integral", C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    ("cmdg_filter_create", C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    ("cmdg_filter_destroy", C.c_int, [_vp, _vp]),
    ("cmdg_filter_apply", C.c_int, [_vp, _vp, _vp, _i32]),
    ("cmdg_set_filters", C.c_int, [_vp, _vp, _vp, _vp]),
    ("cmdg_set_rhs_hooks", C.c_int, [_vp, _vp]),
    ("cmdg_ocean_initialize_states", C.c_int, [_vp, _vp, _vp]),
    ("cmdg_ocean_tendency_from_slow_to_fast", C.c_int, [_vp, _vp, _vp, _vp]),
    ("cmdg_ocean_reconcile_from_fast_to_slow", C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    ("cmdg_split_explicit01_step", C.c_int,
     [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _i32, _vp, _vp, _vp]),
    ("cmdg_group_split_explicit01_step", C.c_int,
     [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _i32, _vp, _vp, _vp]),
    ("cmdg_load_plugin", C.c_int, [C.c_char_p]),
    ("cmdg_lsrk_update", C.c_int, [_vp, _vp, _vp, _d, _d]),
    ("cmdg_ls3n_step", C.c_int, [_vp, _vp, _vp, _vp, _d, _d, _i32, _vp, _vp, _vp]),
    ("cmdg_ssprk_step", C.c_int, [_vp, _vp, _vp, _vp, _d, _d, _i32, _vp, _vp, _vp]),
    ("cmdg_group_split_explicit_step", C.c_int,
     [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _i32, _vp, _vp, _vp]),
    ("cmdg_split_explicit_step", C.c_int,
     [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _i32, _vp, _vp, _vp]),
    ("cmdg_profile_enable", C.c_int, [_vp, _i32]),
    ("cmdg_profile_get", C.c_int, [_vp, _i32, _vp, _vp]),
    ("cmdg_profile_reset", C.c_int, [_vp]),
]

_LIB = None


class CmdgError(RuntimeError):
    pass


def lib():
    """Load ``libcmdg.so`` (after torch, so that both use one HIP runtime)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CmdgError(
                "libcmdg.so is not built (%s); run `python -c 'import __graft_entry__ as g; "
                "g.build()'` -- there is no CPU fallback" % LIB_PATH)
        try:
            import torch  # noqa: F401  (loads libamdhip64 / librccl first)
            # one RCCL instance per process: libcmdg resolves RCCL with dlopen and takes the
            # copy torch.distributed uses (torch/lib/librccl.so) rather than a second one
            rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(rccl):
                os.environ.setdefault("CMDG_RCCL_LIB", rccl)
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)          # AttributeError if a declared symbol is missing
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(status, handle=None):
    if status != 0:
        L = lib()
        msg = L.cmdg_last_error(handle).decode()
        raise CmdgError("libcmdg: %s (%d): %s" % (
            L.cmdg_status_string(status).decode(), status, msg))
