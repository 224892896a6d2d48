"""ctypes binding of libobca_mpc.so (C ABI in include/obca_mpc.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is visible, loading
or solving raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libobca_mpc.so")

OBCA_MAX_OBST = 8
OBCA_MAX_EDGES = 4


class ObcaDims(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("n_obs", ctypes.c_int32), ("m", ctypes.c_int32 * OBCA_MAX_OBST),
                ("max_batch", ctypes.c_int32), ("device", ctypes.c_int32)]


class ObcaWeights(ctypes.Structure):
    _fields_ = [("Q", ctypes.c_double * 9), ("P", ctypes.c_double * 9), ("R1", ctypes.c_double * 4),
                ("R2", ctypes.c_double * 4)]


class ObcaParams(ctypes.Structure):
    """obca_params (include/obca_mpc.h).  A new instance is what obca_params_init leaves: all zero (= every default),
    struct_size set."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("reserved_", ctypes.c_uint32),
                ("free_time", ObcaWeights), ("fixed_time", ObcaWeights),
                ("xL", ctypes.c_double * 2), ("xU", ctypes.c_double * 2),
                ("uL", ctypes.c_double * 2), ("uU", ctypes.c_double * 2),
                ("ego", ctypes.c_double * 4), ("dmin", ctypes.c_double),
                ("tol", ctypes.c_double), ("rho", ctypes.c_double), ("feas_tol", ctypes.c_double),
                ("max_iter_free", ctypes.c_int32), ("max_iter_fixed", ctypes.c_int32), ("max_soc", ctypes.c_int32),
                ("start_order", ctypes.c_int32), ("single_start", ctypes.c_int32), ("patience", ctypes.c_int32),
                ("retry_iter", ctypes.c_int32), ("dodge", ctypes.c_int32), ("terminal_screen", ctypes.c_int32)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = ctypes.sizeof(ObcaParams)


# obca_params.start_order (include/obca_mpc.h)
START_DEFAULT, START_WINDOW_FIRST, START_ZEROS_FIRST, START_X0_FIRST = 0, 1, 2, 3
START_ORDERS = {"default": START_DEFAULT, "x0": START_X0_FIRST, "window": START_WINDOW_FIRST, "zeros": START_ZEROS_FIRST}


class ObcaRolloutDims(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("n_static", ctypes.c_int32), ("m_static", ctypes.c_int32 * OBCA_MAX_OBST),
                ("n_dyn", ctypes.c_int32), ("path_max", ctypes.c_int32), ("batch", ctypes.c_int32),
                ("max_steps", ctypes.c_int32), ("device", ctypes.c_int32), ("N_fix", ctypes.c_int32)]


EXPORTS = ("obca_create", "obca_destroy", "obca_solve_batch", "obca_lds_bytes", "obca_strerror", "obca_version", "obca_params_init",
           "obca_set_profile_buffer", "obca_set_mode", "obca_set_two_sided_sweep", "obca_rollouts_create", "obca_rollouts_destroy", "obca_rollouts_debug_stats", "obca_rollouts_debug_harness",
           "obca_rollouts_reset", "obca_rollouts_step", "obca_rollouts_read", "obca_rollouts_run",
           "obca_rollouts_set_mode", "obca_rollouts_queue_mode", "obca_set_shape_specialisation", "obca_shape_is_specialised", "obca_astar_batch", "obca_astar_workspace_bytes", "obca_primal_size", "obca_set_warm_start",
           "obca_rollouts_set_warm_start", "obca_dual_size", "obca_set_certificate_buffers", "obca_rasterise_batch")

OBCA_MAX_DYN = 4
RUN, DONE_GOAL, DONE_CAP, DONE_FAILED = 0, 1, 2, 3
STATUS_SKIPPED = -5
STATUS_BAD_VARIANT = -6

STATUS_OK, STATUS_ACCEPTABLE, STATUS_INFEASIBLE = 0, 1, 2
STATUS_MAXITER, STATUS_LINESEARCH, STATUS_NUMERIC, STATUS_BAD_BOUNDS = -1, -2, -3, -4

_lib = None


def _bind_hip_runtime():
    """libobca_mpc.so names no HIP runtime of its own (no DT_NEEDED entry, see __graft_entry__.build): its hip* symbols
    bind to the runtime the process already uses, so device memory handed over by the host and our launches always live
    in ONE runtime, whatever the import order.  Under PyTorch-ROCm that is the copy torch ships (torch/lib/libamdhip64.so,
    loaded RTLD_LOCAL by torch -- re-opening the same file RTLD_GLOBAL only widens its visibility); without torch, the
    system runtime.  A C host that links libamdhip64 itself needs nothing of this."""
    cands = []
    try:
        import torch
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except ImportError:
        pass
    cands += ["libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"]
    err = None
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:
            err = e
    raise RuntimeError("no HIP runtime (libamdhip64) could be loaded: %s" % err)


def load():
    """Load the shared library (built by __graft_entry__.build()); raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libobca_mpc.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(expected at %s). There is no CPU fallback." % LIB_PATH)
    _bind_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32p = ctypes.c_void_p, ctypes.c_void_p
    lib.obca_create.argtypes = [ctypes.POINTER(ObcaDims), ctypes.POINTER(ctypes.c_void_p)]
    lib.obca_create.restype = ctypes.c_int
    lib.obca_destroy.argtypes = [ctypes.c_void_p]
    lib.obca_destroy.restype = None
    lib.obca_solve_batch.argtypes = [ctypes.c_void_p, i32p, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp,
                                     ctypes.POINTER(ObcaParams), vp, vp, vp, i32p, i32p, vp, vp]
    lib.obca_solve_batch.restype = ctypes.c_int
    lib.obca_set_profile_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.obca_set_profile_buffer.restype = None
    lib.obca_set_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.obca_set_mode.restype = ctypes.c_int
    lib.obca_set_two_sided_sweep.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.obca_set_two_sided_sweep.restype = ctypes.c_int
    lib.obca_rollouts_create.argtypes = [ctypes.POINTER(ObcaRolloutDims), ctypes.POINTER(ctypes.c_void_p)]
    lib.obca_rollouts_create.restype = ctypes.c_int
    lib.obca_rollouts_destroy.argtypes = [ctypes.c_void_p]
    lib.obca_rollouts_destroy.restype = None
    lib.obca_rollouts_reset.argtypes = [ctypes.c_void_p, vp, vp, vp, vp, vp, vp, vp, ctypes.c_double, ctypes.c_double,
                                        ctypes.POINTER(ObcaParams), vp]
    lib.obca_rollouts_reset.restype = ctypes.c_int
    lib.obca_rollouts_step.argtypes = [ctypes.c_void_p, vp]
    lib.obca_rollouts_step.restype = ctypes.c_int
    lib.obca_rollouts_run.argtypes = [ctypes.c_void_p, ctypes.c_int32, vp]
    lib.obca_rollouts_run.restype = ctypes.c_int
    lib.obca_rollouts_debug_harness.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_double, vp, ctypes.c_int32, vp, vp, vp, vp]
    lib.obca_rollouts_debug_harness.restype = ctypes.c_int
    lib.obca_rollouts_set_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.obca_rollouts_set_mode.restype = ctypes.c_int
    lib.obca_set_shape_specialisation.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.obca_set_shape_specialisation.restype = ctypes.c_int
    lib.obca_shape_is_specialised.argtypes = [ctypes.c_void_p]
    lib.obca_shape_is_specialised.restype = ctypes.c_int
    lib.obca_rollouts_queue_mode.argtypes = [ctypes.c_void_p]
    lib.obca_rollouts_queue_mode.restype = ctypes.c_int
    lib.obca_astar_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.obca_astar_workspace_bytes.restype = ctypes.c_int64
    lib.obca_astar_batch.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp,
                                     ctypes.POINTER(ctypes.c_double), ctypes.c_int32, vp, vp, vp, ctypes.c_int64, vp]
    lib.obca_astar_batch.restype = ctypes.c_int
    lib.obca_rasterise_batch.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, vp, vp]
    lib.obca_rasterise_batch.restype = ctypes.c_int
    lib.obca_primal_size.argtypes = [ctypes.POINTER(ObcaDims)]
    lib.obca_primal_size.restype = ctypes.c_int64
    lib.obca_set_warm_start.argtypes = [ctypes.c_void_p, vp, vp, ctypes.c_double]
    lib.obca_set_warm_start.restype = ctypes.c_int
    lib.obca_dual_size.argtypes = [ctypes.POINTER(ObcaDims)]
    lib.obca_dual_size.restype = ctypes.c_int64
    lib.obca_set_certificate_buffers.argtypes = [ctypes.c_void_p, vp, vp]
    lib.obca_set_certificate_buffers.restype = ctypes.c_int
    lib.obca_rollouts_set_warm_start.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
    lib.obca_rollouts_set_warm_start.restype = ctypes.c_int
    lib.obca_rollouts_read.argtypes = [ctypes.c_void_p] + [vp] * 11
    lib.obca_rollouts_read.restype = ctypes.c_int
    lib.obca_lds_bytes.argtypes = [ctypes.POINTER(ObcaDims)]
    lib.obca_lds_bytes.restype = ctypes.c_int64
    lib.obca_strerror.argtypes = [ctypes.c_int]
    lib.obca_strerror.restype = ctypes.c_char_p
    lib.obca_params_init.argtypes = [ctypes.POINTER(ObcaParams)]
    lib.obca_params_init.restype = None
    lib.obca_version.argtypes = []
    lib.obca_version.restype = ctypes.c_char_p
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise RuntimeError("libobca_mpc: %s (code %d)" % (load().obca_strerror(code).decode(), code))
