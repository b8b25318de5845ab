"""
solver.py -- ctypes binding of libtumnmpc.so with the AcadosOcpSolver method surface.

`BatchedOcpSolver` mirrors the methods of acados_template.AcadosOcpSolver that the reference's
controller classes call (complete list):
    solve, set, get, cost_set, constraints_set, get_cost, get_stats, reset, get_from_qp_in
  call sites: Model_Predictive_Controller/Nominal_NMPC/NMPC_class.py:111-112,172,178,183,193,198,202-205,
  245-246,251,254,295-317; Stochastic_NMPC/SNMPC_class.py:124,130,198;
  Reduced_Robustified_NMPC/Reduced_Robustified_NMPC_class.py:264,280,295,298,335-336,359.
With batch == 1 every method takes / returns exactly the shapes acados does, so the reference's
controller code runs unchanged on it. With batch > 1 values gain a leading batch axis; a value
without it is broadcast to all instances.

There is NO CPU fallback: if the HIP library or a GPU is missing, construction raises.
"""
import contextlib
import ctypes
import os

import numpy as np

from . import config as _config

_HERE = os.path.dirname(os.path.abspath(__file__))
# development build (tests / experiments): the same C-ABI plus the kernel variants "fused" and "pipeline4" (tum_ocp_set_kernel)
DEV_LIB_PATH = os.path.join(_HERE, "libtumnmpc_dev.so")
# overrides: TUM_NMPC_LIB=<path> (A/B testing of builds), TUM_NMPC_DEV=1 (a whole script on the development build)
LIB_PATH = os.environ.get("TUM_NMPC_LIB", DEV_LIB_PATH if os.environ.get("TUM_NMPC_DEV") == "1" else os.path.join(_HERE, "libtumnmpc.so"))
ALL_STAGES = -1


class TumOcpDesc(ctypes.Structure):
    _fields_ = (
        [("N", ctypes.c_int), ("nsub", ctypes.c_int), ("dt", ctypes.c_double),
         ("batch", ctypes.c_int), ("device", ctypes.c_int)]
        + [(n, ctypes.c_double) for n in
           ("lf", "lr", "m", "Iz", "ro", "S", "Cd", "Bf", "Cf", "Df", "Ef", "Br", "Cr", "Dr", "Er",
            "g", "fr0", "fr1", "fr4", "acc_min")]
        + [("n_ggv", ctypes.c_int),
           ("ggv_v", ctypes.c_double * 16), ("ggv_ax", ctypes.c_double * 16), ("ggv_ay", ctypes.c_double * 16),
           ("qp_iter_max", ctypes.c_int),
           ("qp_tol_stat", ctypes.c_double), ("qp_tol_ineq", ctypes.c_double), ("qp_tol_comp", ctypes.c_double),
           ("qp_mu0", ctypes.c_double), ("qp_t0", ctypes.c_double), ("store_qp_in", ctypes.c_int),
           ("qp_warm_start", ctypes.c_int), ("qp_warm_mu", ctypes.c_double),
           ("qp_warm_flips", ctypes.c_int), ("qp_warm_viol", ctypes.c_double)]
    )


_lib = None
_libs = {}              # path -> loaded library
_default_path = None    # None: LIB_PATH


@contextlib.contextmanager
def dev_library():
    """Solvers created inside this context bind to the development build (libtumnmpc_dev.so: the shipped pipeline plus the
    kernel variants "fused" and "pipeline4"). Tests and experiments only."""
    global _default_path
    old = _default_path
    _default_path = DEV_LIB_PATH
    try:
        yield
    finally:
        _default_path = old


# every symbol include/tum_nmpc.h declares (tests/test_cabi.py checks the .so exports them all)
C_SYMBOLS = ["tum_ocp_create", "tum_ocp_free", "tum_ocp_last_error", "tum_ocp_batch", "tum_ocp_horizon",
             "tum_ocp_set", "tum_ocp_get", "tum_ocp_constraints_set", "tum_ocp_cost_set",
             "tum_ocp_solve", "tum_ocp_solve_async", "tum_ocp_synchronize",
             "tum_ocp_get_cost", "tum_ocp_get_stats", "tum_ocp_reset", "tum_ocp_get_from_qp_in",
             "tum_ocp_set_stream", "tum_ocp_get_device", "tum_ocp_put_device", "tum_ocp_bind_device", "tum_ocp_results_async", "tum_ocp_results_wait", "tum_ocp_results_outstanding", "tum_ocp_step_async", "tum_ocp_cold_start", "tum_ocp_last_kernel_ms",
             "tum_ocp_debug_dump", "tum_ocp_profile_phases", "tum_ocp_set_schedule", "tum_ocp_set_kernel",
             "tum_ocp_set_x0_fanout", "tum_pce_moments", "tum_pce_attach", "tum_pce_moments_device",
             "tum_ocp_bounds_snapshot", "tum_ocp_bounds_restore", "tum_ocp_r2_backoff", "tum_ocp_r2_attach", "tum_ocp_constraints_get",
             "tum_ocp_snmpc_attach", "tum_ocp_snmpc_samples", "tum_ocp_snmpc_set_offsets",
             "tum_planner_emulate", "tum_sim_create", "tum_sim_free", "tum_sim_set_state", "tum_sim_plan", "tum_sim_advance",
             "tum_sim_run", "tum_sim_steps", "tum_sim_get", "tum_sim_set_disturbances"]


def load_library(path=None):
    """dlopen libtumnmpc.so (built by __graft_entry__.build()); raises if it is missing."""
    global _lib
    p = path or _default_path or LIB_PATH
    if p in _libs:
        return _libs[p]
    try:
        # torch ships its own HIP runtime: let it load first so that libtumnmpc.so binds to the same
        # libamdhip64 (two runtimes in one process lose the device)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the solver has no CPU fallback)")
    L = ctypes.CDLL(p)
    dp = ctypes.POINTER(ctypes.c_double)
    vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p
    L.tum_ocp_create.restype = vp; L.tum_ocp_create.argtypes = [ctypes.POINTER(TumOcpDesc)]
    L.tum_ocp_free.restype = None; L.tum_ocp_free.argtypes = [vp]
    L.tum_ocp_last_error.restype = cs; L.tum_ocp_last_error.argtypes = []
    L.tum_ocp_batch.argtypes = [vp]; L.tum_ocp_horizon.argtypes = [vp]
    # (the data pointer of the per-call setters / getters goes over as a plain address -- `a.ctypes.data` -- : building a typed ctypes
    #  pointer per call costs 1.9 us, twice the rest of a call; the reference issues 83 of them per control step)
    for name in ("tum_ocp_set", "tum_ocp_constraints_set", "tum_ocp_cost_set"):
        getattr(L, name).argtypes = [vp, ci, cs, vp, ci, ci, ci, ci]
    L.tum_ocp_get.argtypes = [vp, ci, cs, vp, ci, ci, ci, ci]
    L.tum_ocp_get_from_qp_in.argtypes = [vp, ci, cs, dp, ci, ci, ci, ci]
    L.tum_ocp_solve.argtypes = [vp]; L.tum_ocp_solve_async.argtypes = [vp]; L.tum_ocp_synchronize.argtypes = [vp]
    L.tum_ocp_get_cost.argtypes = [vp, dp, ci, ci]
    L.tum_ocp_get_stats.argtypes = [vp, cs, vp, ci, ci]
    L.tum_ocp_reset.argtypes = [vp]; L.tum_ocp_cold_start.argtypes = [vp]
    L.tum_ocp_set_stream.argtypes = [vp, vp]
    L.tum_ocp_set_schedule.argtypes = [vp, ci]
    L.tum_ocp_set_kernel.argtypes = [vp, cs]
    L.tum_ocp_get_device.argtypes = [vp, cs, vp, ci, ci]
    L.tum_ocp_put_device.argtypes = [vp, cs, vp, ci, ci]
    if hasattr(L, "tum_ocp_bind_device"):
        L.tum_ocp_bind_device.argtypes = [vp, cs, vp]
    if hasattr(L, "tum_ocp_results_async"):          # (absent from the libraries of earlier revisions that scripts/dev/ab2.py loads beside this one)
        L.tum_ocp_results_async.argtypes = [vp, ci]
        L.tum_ocp_results_wait.argtypes = [vp, ctypes.POINTER(dp), ctypes.POINTER(dp), ctypes.POINTER(dp)]
    if hasattr(L, "tum_ocp_results_outstanding"):
        L.tum_ocp_results_outstanding.argtypes = [vp]
    if hasattr(L, "tum_ocp_step_async"):
        L.tum_ocp_step_async.argtypes = [vp, vp, vp, ci]
    L.tum_ocp_last_kernel_ms.restype = ctypes.c_double; L.tum_ocp_last_kernel_ms.argtypes = [vp]
    L.tum_ocp_debug_dump.argtypes = [vp, ci, dp, ci]
    L.tum_ocp_profile_phases.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    L.tum_ocp_set_x0_fanout.argtypes = [vp, dp, dp, ci, ci]
    L.tum_pce_moments.argtypes = [vp, cs, ci, dp, ci, ci, dp, dp]
    L.tum_pce_attach.argtypes = [vp, dp, ci, ci]
    L.tum_pce_moments_device.argtypes = [vp, cs, ci, vp, vp]
    L.tum_ocp_bounds_snapshot.argtypes = [vp]; L.tum_ocp_bounds_restore.argtypes = [vp]
    L.tum_ocp_r2_backoff.argtypes = [vp, dp, dp, ci, ctypes.c_double, ctypes.c_double, ctypes.c_double, dp]
    L.tum_ocp_constraints_get.argtypes = [vp, ci, cs, dp, ci, ci]
    L.tum_ocp_r2_attach.argtypes = [vp, dp, dp, ci, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    L.tum_ocp_snmpc_attach.argtypes = [vp, ci, ci, dp, ci, ctypes.c_double]
    L.tum_ocp_snmpc_samples.argtypes = [vp]
    L.tum_ocp_snmpc_set_offsets.argtypes = [vp, dp]
    ip = ctypes.POINTER(ctypes.c_int); cd = ctypes.c_double
    L.tum_planner_emulate.argtypes = [dp, ci, dp, ci, ci, cd, ci, dp, ip, ci]
    L.tum_sim_create.restype = vp; L.tum_sim_create.argtypes = [vp, dp, ci, cd, ci, cd, ci, ip, ci]
    L.tum_sim_free.restype = None; L.tum_sim_free.argtypes = [vp]
    L.tum_sim_set_state.argtypes = [vp, dp, dp, ci]
    L.tum_sim_plan.argtypes = [vp]; L.tum_sim_advance.argtypes = [vp]; L.tum_sim_run.argtypes = [vp, ci]
    L.tum_sim_steps.argtypes = [vp]
    L.tum_sim_get.argtypes = [vp, cs, dp, ctypes.c_longlong]
    if hasattr(L, "tum_sim_set_disturbances"):
        L.tum_sim_set_disturbances.argtypes = [vp, dp, dp, ci]
    _libs[p] = L
    if p == LIB_PATH:
        _lib = L
    return L


def make_desc(N, dt, nsub, batch, device=0, cfg=None, store_qp_in=False, qp_iter_max=50,
              qp_tol=(1e-8, 1e-8, 1e-8), qp_mu0=0.05, qp_t0=0.05, qp_warm_start=True, qp_warm_mu=0.0,
              qp_warm_flips=0, qp_warm_viol=0.0):
    cfg = cfg or _config.default_config()
    d = TumOcpDesc()
    d.N, d.nsub, d.dt, d.batch, d.device = int(N), int(nsub), float(dt), int(batch), int(device)
    for k in ("lf", "lr", "m", "Iz", "ro", "S", "Cd", "acc_min"):
        setattr(d, k, float(cfg["veh"][k]))
    for k in ("Bf", "Cf", "Df", "Ef", "Br", "Cr", "Dr", "Er"):
        setattr(d, k, float(cfg["tire"][k]))
    for k in ("g", "fr0", "fr1", "fr4"):
        setattr(d, k, float(cfg["phys"][k]))
    n = len(cfg["ggv"]["v"])
    if n > 16:
        raise ValueError("ggv table longer than 16 rows")
    d.n_ggv = n
    for i in range(n):
        d.ggv_v[i], d.ggv_ax[i], d.ggv_ay[i] = cfg["ggv"]["v"][i], cfg["ggv"]["ax"][i], cfg["ggv"]["ay"][i]
    d.qp_iter_max = int(qp_iter_max)
    d.qp_tol_stat, d.qp_tol_ineq, d.qp_tol_comp = (float(t) for t in qp_tol)
    d.qp_mu0 = float(qp_mu0)
    d.qp_t0 = float(qp_t0)
    d.store_qp_in = 1 if store_qp_in else 0
    d.qp_warm_start = 1 if qp_warm_start else 0
    d.qp_warm_mu = float(qp_warm_mu)
    d.qp_warm_flips, d.qp_warm_viol = int(qp_warm_flips), float(qp_warm_viol)      # (0: the library's defaults 16 / 0.1)
    return d


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


_FIELD_BYTES = {}          # field name -> bytes, encoded once


class BatchedOcpSolver:
    """`batch` independent copies of the nominal NMPC OCP on one MI355X; acados method names."""

    def __init__(self, N=38, dt=0.08, nsub=3, batch=1, device=0, cfg=None, store_qp_in=False,
                 qp_iter_max=50, qp_tol=(1e-8, 1e-8, 1e-8), qp_mu0=0.05, qp_t0=0.05, qp_warm_start=None, qp_warm_mu=0.0,
                 qp_warm_flips=0, qp_warm_viol=0.0):
        self._L = load_library()
        self.N, self.dt, self.nsub, self.batch = int(N), float(dt), int(nsub), int(batch)
        self.cfg = cfg or _config.default_config()
        if qp_warm_start is None:
            # (acados: qp_solver_warm_start) ON by default for every controller, whichever library is loaded -- a deliberate, documented
            # deviation for the nominal and R2 solvers, where the reference leaves the option at acados' default 0 (INTEGRATION.md,
            # "Interior point warm start"; qp_warm_start=False gives the cold-started method). The development build's extra kernels
            # ("fused", "pipeline4") always cold-start the method and ignore the flag.
            qp_warm_start = True
        self._desc = make_desc(N, dt, nsub, batch, device, self.cfg, store_qp_in, qp_iter_max, qp_tol, qp_mu0, qp_t0, qp_warm_start, qp_warm_mu,
                               qp_warm_flips, qp_warm_viol)
        self._h = self._L.tum_ocp_create(ctypes.byref(self._desc))
        if not self._h:
            raise RuntimeError("tum_ocp_create failed: " + self._err())
        self.status = 0

    # ------------------------------------------------------------------ plumbing
    def _err(self):
        e = self._L.tum_ocp_last_error()
        return e.decode() if e else ""

    def _chk(self, rc, what):
        if rc != 0:
            raise Exception(f"BatchedOcpSolver.{what}: {self._err()}")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.tum_ocp_free(self._h)
                self._h = None
        except Exception:
            pass

    def _put(self, fn, stage, field, value, what):
        v = np.ascontiguousarray(value, dtype=np.float64)
        fb = _FIELD_BYTES.get(field) or _FIELD_BYTES.setdefault(field, field.encode())
        if v.ndim >= 2 and v.shape[0] == self.batch and self.batch > 1:
            v = v.reshape(self.batch, -1)
            ln = v.shape[1]
            self._chk(fn(self._h, stage, fb, v.ctypes.data, ln, 0, self.batch, ln), what)
        else:
            scalar_field = field in ("uh", "lh", "lbu", "ubu") or (field in ("lbx", "ubx") and stage != 0)
            if self.batch > 1 and v.ndim >= 1 and v.shape[0] == self.batch and scalar_field:   # one value per instance
                v = v.reshape(self.batch, 1)
                self._chk(fn(self._h, stage, fb, v.ctypes.data, 1, 0, self.batch, 1), what)
                return
            self._chk(fn(self._h, stage, fb, v.ctypes.data, v.size, 0, self.batch, 0), what)

    def _out(self, a):
        return a[0] if self.batch == 1 else a

    # ------------------------------------------------------------------ acados surface
    def set(self, stage, field, value):
        """acados_solver.set(stage, 'x'|'u'|'yref', value)"""
        self._put(self._L.tum_ocp_set, stage, field, value, "set")

    def get(self, stage, field):
        """acados_solver.get(stage, 'x'|'u'|'sl'|'su')"""
        N = self.N
        if field == "x":
            ln = 8
        elif field == "u":
            ln = 2
        elif field in ("sl", "su"):
            ln = 1 if stage == 0 else (2 if stage == N else 3)
        else:
            raise Exception(f"BatchedOcpSolver.get: unknown field '{field}'")
        out = np.empty((self.batch, ln))
        fb = _FIELD_BYTES.get(field) or _FIELD_BYTES.setdefault(field, field.encode())
        self._chk(self._L.tum_ocp_get(self._h, stage, fb, out.ctypes.data, ln, 0, self.batch, ln), "get")
        return out[0] if self.batch == 1 else out

    def constraints_set(self, stage, field, value):
        self._put(self._L.tum_ocp_constraints_set, stage, field, value, "constraints_set")

    def cost_set(self, stage, field, value):
        """acados_solver.cost_set (NMPC_class.py:294-317). 'W' is PER STAGE, as in acados: cost_set(i, 'W', W) touches stage i only
        (stage N: the 4 x 4 terminal weight); cost_set(ALL_STAGES, 'W', W6x6) sets the stages 0..N-1 in one call. W may be any
        (symmetric) matrix, as in acados: diagonal ones -- all the reference installs -- keep the capsule on the headline's kernels, the first W with
        an off-diagonal entry switches it to the full-W form (include/tum_nmpc.h). 'zl' | 'zu' | 'Zl' | 'Zu': per penalty class (stage 0 / 1..N-1 / N)."""
        v = np.asarray(value, dtype=np.float64)
        if field == "W":
            ny = 6 if (stage < self.N) else 4          # (stage == ALL_STAGES = -1: one 6 x 6 W for all the stages 0..N-1)
            if v.ndim == 3:     # (batch, ny, ny): column-major per instance
                v = np.ascontiguousarray(v.transpose(0, 2, 1)).reshape(v.shape[0], ny * ny)
            else:
                v = np.ascontiguousarray(v.T).reshape(-1)
        self._put(self._L.tum_ocp_cost_set, stage, field, v, "cost_set")

    def solve(self):
        st = self._L.tum_ocp_solve(self._h)
        if st < 0:
            raise Exception("BatchedOcpSolver.solve: " + self._err())
        self.status = st
        return st

    def get_cost(self):
        out = np.zeros(self.batch)
        self._chk(self._L.tum_ocp_get_cost(self._h, _dp(out), 0, self.batch), "get_cost")
        return float(out[0]) if self.batch == 1 else out

    def get_stats(self, field):
        if field in ("time_tot", "time_ipm"):
            o = ctypes.c_double(0.0)
            self._chk(self._L.tum_ocp_get_stats(self._h, field.encode(), ctypes.byref(o), 0, 1), "get_stats")
            return o.value
        if field in ("sqp_iter", "qp_iter", "status", "qp_status"):
            out = np.zeros(self.batch, dtype=np.int32)
            self._chk(self._L.tum_ocp_get_stats(self._h, field.encode(), out.ctypes.data_as(ctypes.c_void_p), 0, self.batch), "get_stats")
            if field == "sqp_iter" and self.batch == 1:
                return int(out[0])
            return out            # acados returns an array for qp_iter; callers take np.max
        if field == "res":
            out = np.zeros((self.batch, 3))
            self._chk(self._L.tum_ocp_get_stats(self._h, b"res", out.ctypes.data_as(ctypes.c_void_p), 0, self.batch), "get_stats")
            return self._out(out)
        raise Exception(f"BatchedOcpSolver.get_stats: unknown field '{field}'")

    def reset(self):
        self._chk(self._L.tum_ocp_reset(self._h), "reset")

    def get_from_qp_in(self, stage, field):
        shp = {"A": (8, 8), "B": (8, 2), "b": (8,)}.get(field)
        if shp is None:
            raise Exception(f"BatchedOcpSolver.get_from_qp_in: unknown field '{field}'")
        n = int(np.prod(shp))
        out = np.zeros((self.batch, n))
        self._chk(self._L.tum_ocp_get_from_qp_in(self._h, stage, field.encode(), _dp(out), n, 0, self.batch, n), "get_from_qp_in")
        if len(shp) == 2:     # the C-ABI hands matrices out column-major (acados convention)
            out = np.ascontiguousarray(out.reshape(self.batch, shp[1], shp[0]).transpose(0, 2, 1))
        return self._out(out)

    # ------------------------------------------------------------------ batch conveniences (no acados counterpart)
    def set_x0(self, x0):
        """lbx_0 = ubx_0 = x0 for every instance; x0: (8,) or (batch, 8)."""
        self.constraints_set(0, "lbx", x0)

    def set_yref_all(self, yref):
        """yref: (N+1, 6) or (batch, N+1, 6); the terminal record uses its first 4 entries."""
        y = np.ascontiguousarray(yref, dtype=np.float64)
        if y.ndim == 3:
            y = y.reshape(y.shape[0], -1)
        else:
            y = y.reshape(-1)
        self._put(self._L.tum_ocp_set, ALL_STAGES, "yref", y, "set_yref_all")

    def set_iterate(self, X=None, U=None):
        if X is not None:
            X = np.ascontiguousarray(X, dtype=np.float64)
            self._put(self._L.tum_ocp_set, ALL_STAGES, "x", X.reshape(X.shape[0], -1) if X.ndim == 3 else X.reshape(-1), "set_iterate")
        if U is not None:
            U = np.ascontiguousarray(U, dtype=np.float64)
            self._put(self._L.tum_ocp_set, ALL_STAGES, "u", U.reshape(U.shape[0], -1) if U.ndim == 3 else U.reshape(-1), "set_iterate")

    def get_iterate(self):
        N = self.N
        X = np.zeros((self.batch, (N + 1) * 8)); U = np.zeros((self.batch, N * 2))
        self._chk(self._L.tum_ocp_get(self._h, ALL_STAGES, b"x", X.ctypes.data, X.shape[1], 0, self.batch, X.shape[1]), "get")
        self._chk(self._L.tum_ocp_get(self._h, ALL_STAGES, b"u", U.ctypes.data, U.shape[1], 0, self.batch, U.shape[1]), "get")
        return X.reshape(self.batch, N + 1, 8), U.reshape(self.batch, N, 2)

    def cold_start(self):
        """X_k = x0 for all k, U = 0 on the device (acados create/reset semantics)."""
        self._chk(self._L.tum_ocp_cold_start(self._h), "cold_start")

    def solve_async(self):
        self._chk(self._L.tum_ocp_solve_async(self._h), "solve_async")

    def synchronize(self):
        self._chk(self._L.tum_ocp_synchronize(self._h), "synchronize")

    def set_stream(self, hip_stream_ptr):
        self._chk(self._L.tum_ocp_set_stream(self._h, ctypes.c_void_p(hip_stream_ptr)), "set_stream")

    def get_device(self, field, dev_ptr, b0=0, nb=None):
        nb = self.batch - b0 if nb is None else nb
        self._chk(self._L.tum_ocp_get_device(self._h, field.encode(), ctypes.c_void_p(dev_ptr), b0, nb), "get_device")

    def put_device(self, field, dev_ptr, b0=0, nb=None):
        """'x0' | 'yref' | 'X' | 'U' from caller-owned device memory (asynchronous D2D on the capsule's stream)"""
        nb = self.batch - b0 if nb is None else nb
        self._chk(self._L.tum_ocp_put_device(self._h, field.encode(), ctypes.c_void_p(dev_ptr), b0, nb), "put_device")

    def bind_device(self, field, dev_ptr):
        """'x0' | 'yref': use the caller's device array (whole batch) IN PLACE as the capsule's own -- no copy; dev_ptr None / 0 hands the
        capsule's own array back. The memory must stay valid and unchanged while solves that use it are in flight."""
        self._chk(self._L.tum_ocp_bind_device(self._h, field.encode(), ctypes.c_void_p(dev_ptr or 0)), "bind_device")

    def results_async(self, with_iterate=False):
        """Enqueue, behind the solve on this capsule's stream, the copy of the results into the capsule's pinned host slabs
        (summary: u0[2], cost, status, qp_iter per instance; with_iterate: also X and U). Returns at once."""
        self._chk(self._L.tum_ocp_results_async(self._h, int(bool(with_iterate))), "results_async")

    def results_wait(self):
        """Block until the copies of the last results_async have landed. Returns numpy VIEWS of the pinned slabs (valid until
        the next results_async on this capsule): (summary (batch, 5), X (batch, N+1, 8) or None, U (batch, N, 2) or None)."""
        dp = ctypes.POINTER(ctypes.c_double)
        ps, px, pu = dp(), dp(), dp()
        self._chk(self._L.tum_ocp_results_wait(self._h, ctypes.byref(ps), ctypes.byref(px), ctypes.byref(pu)), "results_wait")
        B, N = self.batch, self.N
        summ = np.ctypeslib.as_array(ps, shape=(B, 5))
        X = np.ctypeslib.as_array(px, shape=(B, N + 1, 8)) if px else None
        U = np.ctypeslib.as_array(pu, shape=(B, N, 2)) if pu else None
        return summ, X, U

    def step_async(self, x0=None, yref=None, with_iterate=True):
        """One control step of a host-driven loop in one call (tum_ocp_step_async): x0 (batch, 8) and yref (batch, N+1, 6) -- None:
        keep -- uploaded through pinned staging on the capsule's stream, one SQP-RTI behind them, the results request behind the
        solve. Returns at once; results_wait() delivers."""
        def arr(a, n):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.size != n:
                raise Exception(f"BatchedOcpSolver.step: expected {n} values, got {a.size}")
            return a, ctypes.c_void_p(a.ctypes.data)
        a0, p0 = arr(x0, self.batch * 8)
        a1, p1 = arr(yref, self.batch * (self.N + 1) * 6)
        self._chk(self._L.tum_ocp_step_async(self._h, p0, p1, int(bool(with_iterate))), "step_async")

    def results_outstanding(self):
        """result requests enqueued on this capsule and not yet waited for (0, 1 or 2)"""
        if not hasattr(self._L, "tum_ocp_results_outstanding"):      # (a saved build of an earlier round, scripts/dev/ab2.py)
            return 0
        return int(self._L.tum_ocp_results_outstanding(self._h))

    def step(self, x0=None, yref=None, with_iterate=True):
        """step_async + results_wait: (summary (batch, 5): u0[2], cost, status, qp_iter; X; U) as views of the capsule's pinned slabs.
        SYNCHRONOUS: the results returned are those of THIS step. results_wait delivers the OLDEST outstanding request, so requests an
        earlier caller left behind on this capsule (a results_async never waited for, a step whose wait was interrupted) are drained
        and dropped first -- otherwise every later step would hand out the previous step's results, one control step late."""
        while self.results_outstanding() > 0:
            self.results_wait()
        self.step_async(x0, yref, with_iterate)
        return self.results_wait()

    def set_schedule(self, longest_first=True):
        """Dispatch instances longest-first by the previous solve's iteration counts (default) or in natural order."""
        self._chk(self._L.tum_ocp_set_schedule(self._h, int(bool(longest_first))), "set_schedule")

    def set_kernel(self, name):
        """'auto' | 'fused' | 'pipeline' (include/tum_nmpc.h, tum_ocp_set_kernel)"""
        self._chk(self._L.tum_ocp_set_kernel(self._h, name.encode()), "set_kernel")

    def last_kernel_ms(self):
        return float(self._L.tum_ocp_last_kernel_ms(self._h))

    # ------------------------------------------------------------------ K6 / K7
    def set_x0_fanout(self, pose, offsets):
        """x0 of instance p*(S+1)+s = pose[p] (+ offsets[s-1] for s >= 1); batch must be P*(S+1)."""
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(-1, 8)
        offsets = np.ascontiguousarray(offsets, dtype=np.float64).reshape(-1, 8)
        self._chk(self._L.tum_ocp_set_x0_fanout(self._h, _dp(pose), _dp(offsets), pose.shape[0], offsets.shape[0]), "set_x0_fanout")

    def pce_moments(self, field, stage, A):
        """mean / variance over every scenario group of `field` ('x' or 'u') at `stage`; A is the L x S PCE matrix."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        L_, S = A.shape
        m = 8 if field == "x" else 2
        P = self.batch // (S + 1)
        mean = np.zeros((P, m)); var = np.zeros((P, m))
        self._chk(self._L.tum_pce_moments(self._h, field.encode(), stage, _dp(A), L_, S, _dp(mean), _dp(var)), "pce_moments")
        return mean, var

    def pce_attach(self, A):
        """keep the L x S PCE matrix on the device for pce_moments_device"""
        A = np.ascontiguousarray(A, dtype=np.float64)
        self._chk(self._L.tum_pce_attach(self._h, _dp(A), A.shape[0], A.shape[1]), "pce_attach")

    def pce_moments_device(self, field, stage, mean_ptr, var_ptr):
        """asynchronous PCE mean / variance into caller-owned device buffers (P x m doubles each)"""
        self._chk(self._L.tum_pce_moments_device(self._h, field.encode(), int(stage), ctypes.c_void_p(mean_ptr),
                                                 ctypes.c_void_p(var_ptr)), "pce_moments_device")

    def bounds_snapshot(self):
        self._chk(self._L.tum_ocp_bounds_snapshot(self._h), "bounds_snapshot")

    def bounds_restore(self):
        """put back the bounds of the last snapshot (asynchronous, on the capsule's stream)"""
        self._chk(self._L.tum_ocp_bounds_restore(self._h), "bounds_restore")

    def r2_backoff(self, Sigma0, BWB, uph, delta_min, delta_max, uh_nom=1.0, return_backoffs=False):
        """R2NMPC tightening of lbx/ubx/uh for the next solve from the last linearisation (store_qp_in capsules)."""
        S0 = np.ascontiguousarray(Sigma0, dtype=np.float64).reshape(64)
        BW = np.ascontiguousarray(BWB, dtype=np.float64).reshape(64)
        bo = np.zeros((self.batch, self.N, 2)) if return_backoffs else None
        self._chk(self._L.tum_ocp_r2_backoff(self._h, _dp(S0), _dp(BW), int(uph), float(delta_min), float(delta_max),
                                             float(uh_nom), _dp(bo) if bo is not None else None), "r2_backoff")
        return bo

    def r2_attach(self, Sigma0, BWB, uph, delta_min, delta_max, uh_nom=1.0):
        """make the R2NMPC tightening part of every solve (device closed loops); uph = 0 detaches"""
        S = np.ascontiguousarray(Sigma0, dtype=np.float64).reshape(64); B = np.ascontiguousarray(BWB, dtype=np.float64).reshape(64)
        self._chk(self._L.tum_ocp_r2_attach(self._h, _dp(S), _dp(B), int(uph), float(delta_min), float(delta_max), float(uh_nom)), "r2_attach")

    def constraints_get(self, stage, field):
        out = np.zeros(self.batch)
        self._chk(self._L.tum_ocp_constraints_get(self._h, stage, field.encode(), _dp(out), 0, self.batch), "constraints_get")
        return float(out[0]) if self.batch == 1 else out

    def profile_phases(self):
        """One solve with the in-kernel phase timers on: (batch, 12) shader-cycle counters."""
        out = np.zeros((self.batch, 12), dtype=np.int64)
        self._chk(self._L.tum_ocp_profile_phases(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))), "profile_phases")
        return out

    def debug_dump(self, b=0, n=20480):
        out = np.zeros(n)
        self._chk(self._L.tum_ocp_debug_dump(self._h, b, _dp(out), n), "debug_dump")
        return out

    # ------------------------------------------------------------------ reference defaults
    def install_reference_ocp(self, Q=None, R=None, Qe=None, L1=None, L2=None, w_scale=0.01):
        """What NMPC_STM_acados_settings.py:48-60,126-139,161-224 bakes into the solver at creation:
        W = 0.01*blockdiag(Q,R), W_e = 0.01*Qe, bounds on delta_f / steering rate, 0 <= h <= 1,
        slack penalties L1/L2 on every soft bound."""
        mpc, veh = self.cfg["mpc"], self.cfg["veh"]
        if Q is None:
            Q = np.diag([mpc["q_lon"] / mpc["s_lon"] ** 2, mpc["q_lat"] / mpc["s_lat"] ** 2,
                         mpc["q_yaw"] / mpc["s_yaw"] ** 2, mpc["q_vel"] / mpc["s_vel"] ** 2])
        if R is None:
            R = np.diag([mpc["r_jerk"] / mpc["s_jerk"] ** 2, mpc["r_steering_rate"] / mpc["s_steering_rate"] ** 2])
        Qe = Q if Qe is None else Qe
        L1 = mpc["L1_pen"] if L1 is None else L1
        L2 = mpc["L2_pen"] if L2 is None else L2
        N = self.N
        W = np.zeros((6, 6)); W[:4, :4] = Q; W[4:, 4:] = R
        self.cost_set(ALL_STAGES, "W", w_scale * W)          # the same W on every stage < N (per-stage: cost_set(i, "W", ...))
        self.cost_set(N, "W", w_scale * np.asarray(Qe))
        classes = ((0, 1), (1, 3), (N, 2)) if N > 1 else ((0, 1), (N, 2))
        for st, n in classes:                        # one representative stage per penalty class
            for f, val in (("zl", L1), ("zu", L1), ("Zl", L2), ("Zu", L2)):
                self.cost_set(st, f, np.ones(n) * val)
        for k in range(N):
            self.constraints_set(k, "lbu", np.array([veh["delta_f_dot_min"]]))
            self.constraints_set(k, "ubu", np.array([veh["delta_f_dot_max"]]))
        for k in range(1, N + 1):
            self.constraints_set(k, "lbx", np.array([veh["delta_f_min"]]))
            self.constraints_set(k, "ubx", np.array([veh["delta_f_max"]]))
            self.constraints_set(k, "lh", np.array([0.0]))
            self.constraints_set(k, "uh", np.array([1.0]))


class CoupledSnmpcSolver(BatchedOcpSolver):
    """`batch` copies of the reference's coupled SNMPC OCP: what
    Stochastic_NMPC/SNMPC_acados_settings.py:318 builds and SNMPC_class.py:198 solves, acados method names.

    The stacked state is the nominal copy followed by n_s sample copies (nx = 8 (n_s+1)). `Apce` (L x n_s) and `uph`
    stand for the per-stage parameter vector p = [A_pce.flatten(), risk_parameter, stop_flag]; set(stage, "p", ...)
    is accepted and checked against them (the reference re-sends the same p before every solve, SNMPC_class.py:184-193).
    """

    def __init__(self, N=38, dt=0.08, batch=1, Apce=None, uph=5, gamma=0.8, device=0, cfg=None,
                 qp_iter_max=50, qp_tol=(1e-8, 1e-8, 1e-8), qp_mu0=0.05, qp_t0=0.05, x0_offsets=None, qp_warm_start=None, qp_warm_mu=0.0,
                 qp_warm_flips=0, qp_warm_viol=0.0):
        super().__init__(N=N, dt=dt, nsub=1, batch=batch, device=device, cfg=cfg, qp_iter_max=qp_iter_max,
                         qp_tol=qp_tol, qp_mu0=qp_mu0, qp_t0=qp_t0, qp_warm_start=qp_warm_start, qp_warm_mu=qp_warm_mu,
                         qp_warm_flips=qp_warm_flips, qp_warm_viol=qp_warm_viol)
        self.Apce = np.ascontiguousarray(Apce, dtype=np.float64)
        if self.Apce.ndim != 2:
            raise Exception("CoupledSnmpcSolver: Apce must be (num_poly_terms, n_samples)")
        self.L, self.ns = self.Apce.shape
        self.uph, self.gamma = int(uph), float(gamma)
        self.nx = 8 * (self.ns + 1)
        self._chk(self._L.tum_ocp_snmpc_attach(self._h, self.ns, self.L, _dp(self.Apce), self.uph, self.gamma), "snmpc_attach")
        if x0_offsets is not None:
            self.set_x0_offsets(x0_offsets)

    def set_x0_offsets(self, offsets):
        """(n_s, 8) offsets of the sample initial conditions (stds * w_s): afterwards set_x0 / constraints_set(0, 'lbx', x0)
        with the 8 nominal values fan out to the samples on the device (compute_x0dist as a kernel)."""
        o = np.ascontiguousarray(offsets, dtype=np.float64)
        if o.shape != (self.ns, 8):
            raise Exception(f"CoupledSnmpcSolver.set_x0_offsets: expected shape ({self.ns}, 8)")
        self._chk(self._L.tum_ocp_snmpc_set_offsets(self._h, _dp(o)), "set_x0_offsets")

    def set(self, stage, field, value):
        if field == "p":
            # [A_pce.flatten(), risk_parameter, stop_flag] (SNMPC_class.py:124,185,193): handled by the C-ABI -- A_pce and the
            # risk parameter are shared by all stages, the stop flags define the uncertainty propagation horizon of the
            # next solve (include/tum_nmpc.h, tum_ocp_set)
            v = np.ascontiguousarray(value, dtype=np.float64).reshape(-1)
            if v.size != self.L * self.ns + 2:
                raise Exception(f"CoupledSnmpcSolver.set: mismatching dimension for field \"p\" with dimension "
                                f"{self.L * self.ns + 2} (you have {v.size})")
            self._chk(self._L.tum_ocp_set(self._h, int(stage), b"p", v.ctypes.data, v.size, 0, self.batch, 0), "set")
            return
        super().set(stage, field, value)

    def get(self, stage, field):
        if field == "x":
            out = np.zeros((self.batch, self.nx))
            self._chk(self._L.tum_ocp_get(self._h, stage, b"x", out.ctypes.data, self.nx, 0, self.batch, self.nx), "get")
            return self._out(out)
        return super().get(stage, field)

    def get_from_qp_in(self, stage, field):
        raise Exception("CoupledSnmpcSolver.get_from_qp_in: not available for the stacked state")


def planner_emulate(track, poses, n_points, Tp, loop_circuit=True, device=0):
    """Batched PlannerEmulator on the GPU (Utils/MPC_sim_utils.py:137-194): track (n,4) [x,y,yaw,v], poses (P,2).
    Returns (closest_index (P,), ref (P, n_points, 4))."""
    L = load_library()
    track = np.ascontiguousarray(track, dtype=np.float64); poses = np.ascontiguousarray(np.atleast_2d(poses), dtype=np.float64)
    if track.ndim != 2 or track.shape[1] != 4 or poses.shape[1] != 2:
        raise Exception("planner_emulate: track must be (n,4), poses (P,2)")
    P = poses.shape[0]
    ref = np.empty((P, n_points, 4)); idx = np.empty(P, dtype=np.int32)
    rc = L.tum_planner_emulate(_dp(track), track.shape[0], _dp(poses), P, int(n_points), float(Tp), int(bool(loop_circuit)),
                               _dp(ref), idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(device))
    if rc != 0:
        raise Exception("planner_emulate: " + L.tum_ocp_last_error().decode())
    return idx, ref


class DeviceClosedLoop:
    """Closed loops of every instance of a solver kept on the GPU: planner -> SQP-RTI -> plant + estimator
    (main.py:48-78, SimulationMode_main_class.py:106-156). The logs use the reference's npz field names."""

    def __init__(self, solver, track, Tp, Ts=0.02, n_elem=4, windows=(1, 1, 4, 2, 2, 3, 4, 2), loop_circuit=True, log_capacity=0):
        self.solver, self._L = solver, solver._L
        track = np.ascontiguousarray(track, dtype=np.float64)
        win = (ctypes.c_int * 8)(*[int(w) for w in windows])
        self._s = self._L.tum_sim_create(solver._h, _dp(track), track.shape[0], float(Tp), int(bool(loop_circuit)), float(Ts),
                                         int(n_elem), win, int(log_capacity))
        if not self._s:
            raise Exception("tum_sim_create: " + self._L.tum_ocp_last_error().decode())
        self.B = solver.batch
        self.log_capacity = int(log_capacity)

    def __del__(self):
        if getattr(self, "_s", None):
            self._L.tum_sim_free(self._s); self._s = None

    def _chk(self, rc, what):
        if rc != 0:
            raise Exception(f"{what}: " + self._L.tum_ocp_last_error().decode())

    def set_state(self, x_sim, x_mpc, cold_start=True):
        x_sim = np.ascontiguousarray(np.broadcast_to(x_sim, (self.B, 7)), dtype=np.float64)
        x_mpc = np.ascontiguousarray(np.broadcast_to(x_mpc, (self.B, 8)), dtype=np.float64)
        self._chk(self._L.tum_sim_set_state(self._s, _dp(x_sim), _dp(x_mpc), int(cold_start)), "sim_set_state")

    def set_disturbances(self, w_deriv=None, e_est=None):
        """disturbance realisation played back by the loop: (n_steps, B, 7) additive disturbances of the state derivatives and / or
        (n_steps, B, 7) state estimation errors (closed_loop.DisturbanceModel.draw); both None: none"""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (w_deriv, e_est)]
        n = 0
        for a in arrs:
            if a is not None:
                if a.ndim != 3 or a.shape[1:] != (self.B, 7) or (n and a.shape[0] != n):
                    raise Exception("DeviceClosedLoop.set_disturbances: expected (n_steps, batch, 7) arrays of equal length")
                n = a.shape[0]
        self._chk(self._L.tum_sim_set_disturbances(self._s, None if arrs[0] is None else _dp(arrs[0]), None if arrs[1] is None else _dp(arrs[1]), n),
                  "sim_set_disturbances")

    def plan(self):
        self._chk(self._L.tum_sim_plan(self._s), "sim_plan")

    def advance(self):
        self._chk(self._L.tum_sim_advance(self._s), "sim_advance")

    def run(self, nsteps):
        self._chk(self._L.tum_sim_run(self._s, int(nsteps)), "sim_run")

    @property
    def steps(self):
        return self._L.tum_sim_steps(self._s)

    _DIMS = dict(x_sim=7, x_mpc=8, pose=2, ref0=4, closest=1, CiLX=7, MPC_SimX=8, simU=2, simREF=4, simSolverDebug=5)

    def get(self, field):
        d = self._DIMS[field]
        if field in ("x_sim", "x_mpc", "pose", "ref0", "closest"):
            shape = (self.B, d)
        else:
            if self.log_capacity == 0:
                raise Exception(f"DeviceClosedLoop.get('{field}'): created with log_capacity=0 (no logs kept on the device)")
            n = min(self.steps, self.log_capacity) + (1 if field in ("CiLX", "MPC_SimX") else 0)   # the first log_capacity steps
            shape = (n, self.B, d)
        out = np.empty(shape)
        self._chk(self._L.tum_sim_get(self._s, field.encode(), _dp(out), out.size), "sim_get " + field)
        return out[:, 0].astype(np.int64) if field == "closest" else out

    @property
    def graph_steps(self):
        out = np.zeros(1)
        self._chk(self._L.tum_sim_get(self._s, b"graph_steps", _dp(out), 1), "sim_get graph_steps")
        return int(out[0])

    def logs(self):
        return {k: self.get(k) for k in ("CiLX", "MPC_SimX", "simU", "simREF", "simSolverDebug")}
