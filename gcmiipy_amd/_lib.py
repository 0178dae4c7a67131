"""ctypes binding of libgcmcore.so (include/gcmcore.h).

The library is the product: if it is missing or exports fewer symbols than the
header declares, importing this module raises -- there is no NumPy fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCMCORE_LIB", os.path.join(_HERE, "lib", "libgcmcore.so"))

ABI_VERSION = 1

# enums of include/gcmcore.h
SW2D, SW2D_TEMP, PE2D, PE25D = 1, 2, 3, 4
P, U, V, T, Q = 0, 1, 2, 3, 4
TRACER_NONE, TRACER_UPWIND, TRACER_VANLEER = 0, 1, 2
VARIANT_AUTO, VARIANT_STAGED, VARIANT_FUSED = 0, 1, 2
F64, F32 = 0, 1
ADV_UPWIND, ADV_FV_UPWIND, ADV_FV_PLAIN, ADV_VANLEER, ADV_MOMENTUM = range(5)
DIAG_ANY_NAN, DIAG_MAX_U, DIAG_MEAN_P, DIAG_SUM_P, DIAG_MIN_U, DIAG_MAX_V, DIAG_MIN_V = range(7)
DIAG_TV_P, DIAG_TV_U, DIAG_TV_V, DIAG_TV_T, DIAG_TV_Q = range(7, 12)
INT_SPU, INT_PIT, INT_PN, INT_PHI, INT_PGFU = range(5)
FL_VAN_LEER, FL_CALC_R, FL_DONOR_FLUX, FL_DONOR_ADVECTION = range(4)
OP1D_ADVEC_Q, OP1D_CALC_PU, OP1D_UN_PU, OP1D_ADVEC_P, OP1D_ADVEC_PU, OP1D_ADVEC_T, OP1D_PGF = range(7)
(PEOP_CALC_PU, PEOP_CALC_PV, PEOP_UN_PU, PEOP_UN_PV, PEOP_AFLUX, PEOP_ADVEC_SIG, PEOP_ADVEC_M_PU, PEOP_GEOPOTENTIAL,
 PEOP_PGF, PEOP_ADVEC_T) = range(10)


class PeGeom(C.Structure):
    """gcm_pe_geom"""
    _fields_ = [("dx_j", C.c_void_p), ("dx_h", C.c_void_p), ("dsig", C.c_void_p), ("sig", C.c_void_p),
                ("sigb", C.c_void_p), ("sigt", C.c_void_p), ("heightmap", C.c_void_p), ("dy", C.c_double),
                ("ptop", C.c_double)]


(OP_ADV_U, OP_ADV_V, OP_GEO_GRAD_U, OP_GEO_GRAD_V, OP_ADV_GEO, OP_LAPLACIAN, OP_VISCOSITY, OP_DENSITY_FROM,
 OP_GEOPOTENTIAL_FROM, OP_TO_TRUE_TEMP, OP_TO_POTENTIAL_TEMP, OP_TO_DENSITY, OP_SCALING, OP_UNSCALING,
 OP_PE2D_ADVEC_P, OP_PE2D_DUT, OP_PE2D_DVT, OP_PE2D_PGF_U, OP_PE2D_PGF_V) = range(19)
OK, ERR_ARG, ERR_HIP, ERR_NODEVICE, ERR_STATE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5

_dp = C.POINTER(C.c_double)


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("model", C.c_int32), ("width", C.c_int32),
        ("height", C.c_int32), ("layers", C.c_int32), ("tracer", C.c_int32),
        ("variant", C.c_int32), ("filter", C.c_int32), ("nranks", C.c_int32),
        ("rank", C.c_int32), ("global_height", C.c_int32), ("row0", C.c_int32),
        ("device", C.c_int32), ("dtype", C.c_int32), ("halo_steps", C.c_int32),
        ("dx", C.c_double), ("dy", C.c_double), ("ptop", C.c_double),
        ("dx_j", _dp), ("dx_h", _dp), ("sig", _dp), ("dsig", _dp), ("sigb", _dp),
        ("sigt", _dp), ("heightmap", _dp), ("cor_u", _dp), ("cor_v", _dp), ("stream", C.c_void_p),
    ]


class Exchange(C.Structure):
    """gcm_exchange of include/gcmcore.h"""
    _fields_ = [("comm", C.c_void_p), ("north", C.c_int32), ("south", C.c_int32),
                ("send", C.c_void_p), ("recv", C.c_void_p), ("group_start", C.c_void_p), ("group_end", C.c_void_p),
                ("send_north", C.c_void_p), ("send_south", C.c_void_p), ("recv_north", C.c_void_p),
                ("recv_south", C.c_void_p)]


class Physics(C.Structure):
    """gcm_physics of include/gcmcore.h"""
    _fields_ = [("utc", C.c_double), ("t_lw", C.c_double), ("t_sw", C.c_double), ("albedo", C.c_double),
                ("lat", _dp), ("lon", _dp)]


_H = C.c_void_p
# name -> (restype, argtypes); every symbol include/gcmcore.h declares
SYMBOLS = {
    "gcm_abi_version": (C.c_int, []),
    "gcm_device_count": (C.c_int, []),
    "gcm_build_info": (C.c_char_p, []),
    "gcm_exner_table": (C.c_int, [_dp]),
    "gcm_filter_plan": (C.c_int, [C.c_int, C.POINTER(C.c_uint), C.c_int]),
    "gcm_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "gcm_destroy": (C.c_int, [_H]),
    "gcm_last_error": (C.c_char_p, [_H]),
    "gcm_set_state": (C.c_int, [_H] + [C.c_void_p] * 5),
    "gcm_get_state": (C.c_int, [_H] + [C.c_void_p] * 5),
    "gcm_step": (C.c_int, [_H, C.c_int, C.c_double]),
    "gcm_half_step": (C.c_int, [_H, C.c_int, C.c_double]),
    "gcm_get_star": (C.c_int, [_H] + [C.c_void_p] * 5),
    "gcm_set_star": (C.c_int, [_H] + [C.c_void_p] * 5),
    "gcm_diag": (C.c_int, [_H, C.c_int, _dp]),
    "gcm_energy": (C.c_int, [_H, _dp, C.c_int, _dp]),
    "gcm_stats": (C.c_int, [_H, _dp, C.c_int, _dp]),
    "gcm_flux_limiter": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]),
    "gcm_polar_filter": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "gcm_get_intermediate": (C.c_int, [_H, C.c_int, C.c_void_p]),
    "gcm_set_ground": (C.c_int, [_H, C.c_void_p]),
    "gcm_get_ground": (C.c_int, [_H, C.c_void_p]),
    "gcm_grey_radiation": (C.c_int, [_H] + [C.c_double] * 4 + [_dp, _dp, C.c_void_p, C.c_void_p]),
    "gcm_solar_step": (C.c_int, [_H] + [C.c_double] * 5 + [_dp, _dp]),
    "gcm_set_physics": (C.c_int, [_H, C.c_void_p]),
    "gcm_get_utc": (C.c_int, [_H, _dp]),
    "gcm_snapshot": (C.c_int, [_H]),
    "gcm_restore": (C.c_int, [_H]),
    "gcm_halo_bytes": (C.c_size_t, [_H]),
    "gcm_halo_pack": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "gcm_halo_unpack": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "gcm_halo_pack2": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gcm_set_halo_buffers": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "gcm_wait_edges": (C.c_int, [_H, C.c_void_p]),
    "gcm_comm_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "gcm_halo_unpack2": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gcm_step_interior": (C.c_int, [_H, C.c_double, C.c_void_p]),
    "gcm_step_boundary": (C.c_int, [_H, C.c_double, C.c_void_p]),
    "gcm_step_phase": (C.c_int, [_H, C.c_int, C.c_double, C.c_void_p]),
    "gcm_set_exchange": (C.c_int, [_H, C.POINTER(Exchange)]),
    "gcm_band_run": (C.c_int, [_H, C.c_int, C.c_double]),
    "gcm_set_band_overlap": (C.c_int, [_H, C.c_int]),
    "gcm_sync": (C.c_int, [_H]),
    "gcm_advect2d": (C.c_int, [C.c_int] * 6 + [C.c_double] * 3 + [C.c_void_p] * 3),
    "gcm_pgf2d": (C.c_int, [C.c_int] * 3 + [C.c_double] * 3 + [C.c_void_p] * 3),
    "gcm_pe1d": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p * 4),
                           C.POINTER(C.c_void_p * 4), C.POINTER(C.c_void_p * 4)]),
    "gcm_sw2d_op": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p]),
    "gcm_pe25d_op": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p * 5),
                               C.POINTER(C.c_void_p * 4)]),
    "gcm_pe25d_op_last_error": (C.c_char_p, []),
    "gcm_pe1d_op": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gcm_array_stats": (C.c_int, [C.c_void_p, C.c_long, C.c_long, C.c_void_p]),
    "gcm_ops_last_error": (C.c_char_p, []),
    "gcm_ops_release_scratch": (C.c_int, []),
    "gcm_time_steps": (C.c_int, [_H, C.c_int, C.c_double, _dp, _dp]),
}


def _preload_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, but requested by
    file name).  If the system runtime is loaded first and torch later, the process ends up with
    TWO HIP runtimes and torch.cuda fails with "No HIP GPUs are available".  Loading torch's copy
    first (without importing torch) makes libgcmcore.so and torch share one runtime in either
    import order.  GCMCORE_SYSTEM_HIP=1 skips this."""
    if os.environ.get("GCMCORE_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def _load():
    _preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gcmiipy_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C gcmiipy_amd/csrc`; there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError("gcmiipy_amd: %s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.gcm_abi_version() != ABI_VERSION:
        raise ImportError("gcmiipy_amd: ABI version mismatch")
    return lib


lib = _load()
