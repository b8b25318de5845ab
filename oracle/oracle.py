"""
oracle/oracle.py -- ctypes wrapper around oracle/liboracle.so.

TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this module; the product (tum-control_amd/) never does.

The C file restates the acados SQP-RTI step the reference calls at
Model_Predictive_Controller/Nominal_NMPC/NMPC_class.py:183; see its header for the
full list of reference lines followed. Parity pin: tests/test_oracle_golden.py checks it
against the reference's logged acados outputs committed under tests/golden/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class StmModel(ctypes.Structure):
    _fields_ = (
        [(n, ctypes.c_double) for n in
         ("lf", "lr", "m", "Iz", "ro", "S", "Cd",
          "Bf", "Cf", "Df", "Ef", "Br", "Cr", "Dr", "Er",
          "g", "fr0", "fr1", "fr4", "acc_min")]
        + [("n_ggv", ctypes.c_int),
           ("ggv_v", ctypes.c_double * 16), ("ggv_ax", ctypes.c_double * 16), ("ggv_ay", ctypes.c_double * 16)]
    )


# EDGAR constants, restated from the reference's config data:
#   Config/EDGAR/veh_params_pred.yaml:3-10,16-25, Config/EDGAR/pacejka_params.yaml:3-12,
#   Config/EDGAR/ggv.csv, Prediction_Models/pred_model_dynamic_stm_pacejka.py:38-46
EDGAR = dict(
    lf=1.484, lr=1.644, m=2520.0, Iz=13600.0, ro=1.225, S=2.9, Cd=0.35,
    Bf=10.0, Cf=1.3, Df=15591.427, Ef=0.97, Br=10.0, Cr=1.6, Dr=24629.523, Er=0.97,
    g=9.81, fr0=0.009, fr1=0.002, fr4=0.0003, acc_min=-3.5,
    ggv_v=[0, 4, 8, 11.11, 12, 20, 24, 28, 32, 37.5],
    ggv_ax=[3, 3, 3, 3, 2.5, 2.5, 2.5, 2.5, 2.5, 2.5],
    ggv_ay=[5.886] * 10,
    delta_f_min=-0.610865, delta_f_max=0.610865,
    delta_f_dot_min=-0.322, delta_f_dot_max=0.322,
)


def edgar_model():
    m = StmModel()
    for k in ("lf", "lr", "m", "Iz", "ro", "S", "Cd", "Bf", "Cf", "Df", "Ef", "Br", "Cr", "Dr", "Er",
              "g", "fr0", "fr1", "fr4", "acc_min"):
        setattr(m, k, EDGAR[k])
    n = len(EDGAR["ggv_v"])
    m.n_ggv = n
    for i in range(n):
        m.ggv_v[i] = EDGAR["ggv_v"][i]
        m.ggv_ax[i] = EDGAR["ggv_ax"][i]
        m.ggv_ay[i] = EDGAR["ggv_ay"][i]
    return m


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "nmpc_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def use_native():
    """bench.py's cpu_baseline leg: rebuild the checker with -march=native on THIS host (oracle/_native/, next to the shipped
    x86-64-v3 build) and switch this module to it. Returns a description of the build in use; falls back to the shipped
    library when no compiler is present."""
    global _lib, _LIB_PATH
    try:
        out_dir = os.path.join(_HERE, "_native")
        os.makedirs(out_dir, exist_ok=True)
        out = os.path.join(out_dir, "liboracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o", out,
                               os.path.join(_HERE, "nmpc_oracle.c"), "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _LIB_PATH, _lib = out, None
        lib()
        return "gcc -O3 -march=native -fopenmp, rebuilt on this host"
    except Exception:
        _LIB_PATH, _lib = os.path.join(_HERE, "liboracle.so"), None
        return "gcc -O3 -march=x86-64-v3 -fopenmp (shipped build; no compiler on this host)"


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        L.oracle_create.restype = ctypes.c_void_p
        L.oracle_create.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int]
        L.oracle_free.argtypes = [ctypes.c_void_p]
        L.oracle_field.restype = dp
        L.oracle_field.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.oracle_set_model.argtypes = [ctypes.c_void_p, ctypes.POINTER(StmModel)]
        L.oracle_set_iter_max.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_solve.argtypes = [ctypes.c_void_p]
        L.oracle_solve.restype = ctypes.c_int
        L.oracle_qp_iter.argtypes = [ctypes.c_void_p]
        L.oracle_status.argtypes = [ctypes.c_void_p]
        L.oracle_stm_f.argtypes = [ctypes.POINTER(StmModel), dp, dp, dp, dp, dp]
        L.oracle_rk4_sens.argtypes = [ctypes.POINTER(StmModel), dp, dp, ctypes.c_double, ctypes.c_int, dp, dp, dp]
        L.oracle_h.argtypes = [ctypes.POINTER(StmModel), dp, dp, dp]
        L.oracle_set_debug.argtypes = [dp]
        L.oracle_solve_batch_cold.argtypes = [ctypes.c_void_p, ctypes.c_int, dp, dp, dp, dp, dp, ctypes.c_int]
        L.oracle_solve_batch_cold_forced.argtypes = [ctypes.c_void_p, ctypes.c_int, dp, dp, dp, dp, dp, ctypes.c_int,
                                                     ctypes.POINTER(ctypes.c_int)]
        L.oracle_set_iter_force.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_forget_qp.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def stm_f(x, u, model=None):
    """xdot, Jx (8x8), Ju (8x2) of the single-track model."""
    model = model or edgar_model()
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    xd = np.zeros(8); Jx = np.zeros((8, 8)); Ju = np.zeros((8, 2))
    lib().oracle_stm_f(ctypes.byref(model), _dp(x), _dp(u), _dp(xd), _dp(Jx), _dp(Ju))
    return xd, Jx, Ju


def rk4_sens(x, u, dt, nsub, model=None):
    model = model or edgar_model()
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    xn = np.zeros(8); A = np.zeros((8, 8)); B = np.zeros((8, 2))
    lib().oracle_rk4_sens(ctypes.byref(model), _dp(x), _dp(u), float(dt), int(nsub), _dp(xn), _dp(A), _dp(B))
    return xn, A, B


def h_con(x, model=None):
    model = model or edgar_model()
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.zeros(1); gh = np.zeros(8)
    lib().oracle_h(ctypes.byref(model), _dp(x), _dp(h), _dp(gh))
    return float(h[0]), gh


class OracleOcp:
    """One nominal-NMPC OCP instance (N stages), CPU FP64.

    Array attributes are numpy views straight into the C struct:
      X (N+1,8)  U (N,2)  x0 (8)  yref (N+1,6)  W (N+1,6: diagonal of W per stage)
      lbu/ubu (N)  lbx/ubx (N+1)  lh/uh (N+1)  zl/zu/Zl/Zu (N+1,3: slots bu,bx,h)
    """

    def __init__(self, N=38, dt=0.08, nsub=3, model=None):
        L = lib()
        self._h = L.oracle_create(int(N), float(dt), int(nsub))
        if not self._h:
            raise ValueError("bad horizon")
        self.N, self.dt, self.nsub = N, dt, nsub
        self._model = model or edgar_model()
        L.oracle_set_model(self._h, ctypes.byref(self._model))
        self.X = self._view("X").reshape(N + 1, 8)
        self.U = self._view("U").reshape(N, 2)
        self.x0 = self._view("x0")
        self.yref = self._view("yref").reshape(N + 1, 6)
        self.W = self._view("W").reshape(N + 1, 6)
        self.Wf = self._view("Wf").reshape(N + 1, 6, 6)          # full W per stage (stage N: its leading 4 x 4), used when full_w[0] != 0
        self.full_w = self._view("full_w")
        self.lbu, self.ubu = self._view("lbu"), self._view("ubu")
        self.lbx, self.ubx = self._view("lbx"), self._view("ubx")
        self.lh, self.uh = self._view("lh"), self._view("uh")
        self.zl, self.zu = self._view("zl").reshape(N + 1, 3), self._view("zu").reshape(N + 1, 3)
        self.Zl, self.Zu = self._view("Zl").reshape(N + 1, 3), self._view("Zu").reshape(N + 1, 3)
        self.A = self._view("A").reshape(N, 8, 8)
        self.B = self._view("B").reshape(N, 8, 2)
        self.b = self._view("b").reshape(N, 8)
        self.sl, self.su = self._view("sl"), self._view("su")
        self.ipm_tol = self._view("ipm_tol")
        self.ipm_mu0 = self._view("ipm_mu0")
        self.ipm_t0 = self._view("ipm_t0")
        self.ipm_reg = self._view("ipm_reg")
        self.res = self._view("res")
        # bounds of the shipped OCP (NMPC_STM_acados_settings.py:108-139)
        self.lbu[:] = EDGAR["delta_f_dot_min"]; self.ubu[:] = EDGAR["delta_f_dot_max"]
        self.lbx[:] = EDGAR["delta_f_min"]; self.ubx[:] = EDGAR["delta_f_max"]
        self.lh[:] = 0.0; self.uh[:] = 1.0

    def _view(self, name):
        n = ctypes.c_int(0)
        p = lib().oracle_field(self._h, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def __del__(self):
        try:
            if self._h:
                lib().oracle_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_weights(self, q_xy, q_yaw, q_vel, r_jerk, r_steer, L1, L2, scale=1.0):
        """update_cost_function_weights (NMPC_class.py:269-317): W = blockdiag(Q,R) raw."""
        self.W[:] = scale * np.array([q_xy, q_xy, q_yaw, q_vel, r_jerk, r_steer])
        for a in (self.zl, self.zu):
            a[:] = L1
        for a in (self.Zl, self.Zu):
            a[:] = L2

    def set_full_W(self, W):
        """cost_set(i, 'W', W) with an arbitrary symmetric W: (N+1, 6, 6) (stage N: the leading 4 x 4 is used) or one (6, 6) for every stage.
        None: back to the diagonal self.W."""
        if W is None:
            self.full_w[0] = 0.0
            return
        W = np.asarray(W, dtype=float)
        self.Wf[:] = 0.5 * (W + np.swapaxes(W, -1, -2))
        self.full_w[0] = 1.0

    def cold_start(self, x0):
        """acados create / reset(): x_k = x0 for all k, u = 0 (NMPC_class.py:250-254)."""
        self.x0[:] = x0
        self.X[:] = np.asarray(x0)[None, :]
        self.U[:] = 0.0
        lib().oracle_forget_qp(self._h)          # (the interior point method of the next solve starts cold too)

    def qp_warm_start(self, on=True, warm_mu=0.0):
        """interior point warm start from the previous QP's multipliers in a sequence of solves (default: on, IPM_WARM_DEFAULT)"""
        self.set_ipm_experiment(5 if on else 0, warm_mu, 0, 0.0)

    def set_yref(self, pos_x, pos_y, ref_yaw, ref_v):
        """NMPC_class.py:169-180"""
        n = self.N + 1
        self.yref[:] = 0.0
        self.yref[:, 0] = pos_x[:n]; self.yref[:, 1] = pos_y[:n]
        self.yref[:, 2] = ref_yaw[:n]; self.yref[:, 3] = ref_v[:n]

    def set_iter_max(self, it):
        lib().oracle_set_iter_max(self._h, int(it))

    def set_iter_force(self, it):
        """tests: run exactly `it` interior point iterations (0 = normal termination test)"""
        lib().oracle_set_iter_force(self._h, int(it))

    def set_ipm_experiment(self, warm=0, warm_mu=0.0, ncorr=0, dalpha=0.0):
        """iteration-count experiments of the interior point method (scripts/study/ipm_iterations.py; all 0 = the shipped method):
        warm-start variant and its complementarity target, number of Gondzio centrality correctors and their step enlargement"""
        L = lib()
        L.oracle_set_ipm_experiment.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double]
        L.oracle_set_ipm_experiment(self._h, int(warm), float(warm_mu), int(ncorr), float(dalpha))

    def set_ipm_split(self, split=0):
        """same study: separate primal / dual step lengths"""
        L = lib()
        L.oracle_set_ipm_split.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_set_ipm_split(self._h, int(split))

    def set_ipm_skip(self, sigma_thr=0.0, amax_thr=0.0):
        """study switch: skip the corrector pass when the predictor asks for centring sigma <= sigma_thr at a step to the boundary >= amax_thr"""
        L = lib()
        L.oracle_set_ipm_skip.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
        L.oracle_set_ipm_skip(self._h, float(sigma_thr), float(amax_thr))

    def set_ipm_vstart(self, vstart=0, q_threshold=0.0):
        """same study: primal start at the unconstrained minimiser (1: always, 2: only if |q|_inf > q_threshold)"""
        L = lib()
        L.oracle_set_ipm_vstart.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
        L.oracle_set_ipm_vstart(self._h, int(vstart), float(q_threshold))

    def solve(self):
        return lib().oracle_solve(self._h)

    def solve_debug(self):
        """solve() that also returns the condensed QP (H, q, C, d, g) it built."""
        N = self.N; nv, m = 2 * N, 3 * N
        buf = np.zeros(nv * nv + nv + m * nv + m + (N + 1) * 8)
        lib().oracle_set_debug(_dp(buf))
        try:
            st = lib().oracle_solve(self._h)
        finally:
            lib().oracle_set_debug(None)
        o = 0
        H = buf[o:o + nv * nv].reshape(nv, nv); o += nv * nv
        q = buf[o:o + nv]; o += nv
        C = buf[o:o + m * nv].reshape(m, nv); o += m * nv
        d = buf[o:o + m]; o += m
        g = buf[o:].reshape(N + 1, 8)
        return st, dict(H=H, q=q, C=C, d=d, g=g)

    @property
    def cost(self):
        return float(self._view("cost")[0])

    @property
    def qp_iter(self):
        return lib().oracle_qp_iter(self._h)

    @property
    def status(self):
        return lib().oracle_status(self._h)

    def solve_batch_cold(self, x0, yref, nthreads=1, force_iter=None):
        """cpu_baseline helper: nb independent cold-start solves sharing this OCP's data. force_iter (tests): interior point
        iteration count imposed per instance."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        yref = np.ascontiguousarray(yref, dtype=np.float64)
        nb = x0.shape[0]
        assert yref.shape == (nb, self.N + 1, 6)
        u0 = np.zeros((nb, 2)); X1 = np.zeros((nb, 8)); stats = np.zeros((nb, 3))
        if force_iter is None:
            lib().oracle_solve_batch_cold(self._h, nb, _dp(x0), _dp(yref), _dp(u0), _dp(X1), _dp(stats), int(nthreads))
        else:
            f = np.ascontiguousarray(force_iter, dtype=np.int32)
            assert f.shape == (nb,)
            lib().oracle_solve_batch_cold_forced(self._h, nb, _dp(x0), _dp(yref), _dp(u0), _dp(X1), _dp(stats), int(nthreads),
                                                 f.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        return u0, X1, stats


def _snmpc_bind(L):
    if getattr(L, "_snmpc_bound", False):
        return L
    dp = ctypes.POINTER(ctypes.c_double)
    L.snmpc_create.restype = ctypes.c_void_p
    L.snmpc_create.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    L.snmpc_free.argtypes = [ctypes.c_void_p]
    L.snmpc_field.restype = dp
    L.snmpc_field.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.snmpc_set_model.argtypes = [ctypes.c_void_p, ctypes.POINTER(StmModel)]
    L.snmpc_set_iter_max.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.snmpc_forget_qp.argtypes = [ctypes.c_void_p]
    L.snmpc_solve.argtypes = [ctypes.c_void_p]
    L.snmpc_solve.restype = ctypes.c_int
    L.snmpc_qp_iter.argtypes = [ctypes.c_void_p]
    L.snmpc_status.argtypes = [ctypes.c_void_p]
    L.snmpc_set_debug.argtypes = [dp]
    L.oracle_h_vabs.argtypes = [ctypes.POINTER(StmModel), dp, dp, dp]
    L.snmpc_eval.argtypes = [ctypes.c_void_p, ctypes.c_double, dp, dp, dp, dp, dp]
    L._snmpc_bound = True
    return L


def h_con_vabs(x, model=None):
    """gg circle with the limits looked up at |v| (SNMPC_acados_settings.py:60-67)."""
    model = model or edgar_model()
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.zeros(1); gh = np.zeros(8)
    _snmpc_bind(lib()).oracle_h_vabs(ctypes.byref(model), _dp(x), _dp(h), _dp(gh))
    return float(h[0]), gh


class OracleSnmpcOcp:
    """The coupled SNMPC OCP (nominal copy + n_s sample copies, one shared input), CPU FP64.

    Restates `acados_solver.solve()` at Stochastic_NMPC/SNMPC_class.py:198 (see the C file's SNMPC section).
    X (N+1, n_s+1, 8); x0 (n_s+1, 8); Apce (L, n_s); stop (N+1); the rest as OracleOcp.
    """

    def __init__(self, N=38, dt=0.08, Apce=None, uph=5, gamma=0.8, model=None):
        L = _snmpc_bind(lib())
        Apce = np.ascontiguousarray(Apce, dtype=np.float64)
        self.L, self.ns = Apce.shape
        self._h = L.snmpc_create(int(N), float(dt), int(self.ns), int(self.L), float(gamma))
        if not self._h:
            raise ValueError("bad dimensions")
        self.N, self.dt = N, dt
        self._model = model or edgar_model()
        L.snmpc_set_model(self._h, ctypes.byref(self._model))
        ns = self.ns
        self.X = self._view("X").reshape(N + 1, ns + 1, 8)
        self.U = self._view("U").reshape(N, 2)
        self.x0 = self._view("x0").reshape(ns + 1, 8)
        self.yref = self._view("yref").reshape(N + 1, 6)
        self.W = self._view("W").reshape(N + 1, 6)
        self.lbu, self.ubu = self._view("lbu"), self._view("ubu")
        self.lbx, self.ubx = self._view("lbx"), self._view("ubx")
        self.lh, self.uh = self._view("lh"), self._view("uh")
        self.zl, self.zu = self._view("zl").reshape(N + 1, 3), self._view("zu").reshape(N + 1, 3)
        self.Zl, self.Zu = self._view("Zl").reshape(N + 1, 3), self._view("Zu").reshape(N + 1, 3)
        self.Apce = self._view("Apce").reshape(self.L, ns)
        self.stop = self._view("stop")
        self.sl, self.su = self._view("sl"), self._view("su")
        self.hval = self._view("hval")
        self.ipm_tol = self._view("ipm_tol")
        self.res = self._view("res")
        self.Apce[:] = Apce
        # stop flags of the uncertainty propagation horizon (SNMPC_class.py:103-104)
        self.stop[:] = 0.0
        self.stop[uph:] = 1.0
        self.lbu[:] = EDGAR["delta_f_dot_min"]; self.ubu[:] = EDGAR["delta_f_dot_max"]
        self.lbx[:] = EDGAR["delta_f_min"]; self.ubx[:] = EDGAR["delta_f_max"]
        self.lh[:] = 0.0; self.uh[:] = 1.0

    def _view(self, name):
        n = ctypes.c_int(0)
        p = lib().snmpc_field(self._h, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def __del__(self):
        try:
            if self._h:
                lib().snmpc_free(self._h)
                self._h = None
        except Exception:
            pass

    set_weights = OracleOcp.set_weights
    set_yref = OracleOcp.set_yref

    def cold_start(self, x0_samples):
        """SNMPC_class.py:119-127: lbx_0 = ubx_0 = x0_samples, x_k = x0_samples for all k, u = 0."""
        self.x0[:] = x0_samples
        self.X[:] = np.asarray(x0_samples)[None]
        self.U[:] = 0.0
        lib().snmpc_forget_qp(self._h)

    def qp_warm_start(self, on=True, warm_mu=0.0):
        """interior point warm start from the previous QP (default: on, as the reference's SNMPC solver: qp_solver_warm_start = 1)"""
        L = lib()
        L.snmpc_set_ipm_warm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
        L.snmpc_set_ipm_warm(self._h, 5 if on else 0, float(warm_mu))

    def set_initial_state(self, x0_samples):
        self.x0[:] = x0_samples

    def set_iter_max(self, it):
        lib().snmpc_set_iter_max(self._h, int(it))

    def solve(self):
        return lib().snmpc_solve(self._h)

    def eval_stage(self, x, u, stop):
        """model functions at one point: (f_disc(x,u,p), cost_y_expr(x,u), con_h_expr(x,p)) for a stage with this stop_flag"""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1); u = np.ascontiguousarray(u, dtype=np.float64)
        xn = np.zeros_like(x); y = np.zeros(6); h = np.zeros(1)
        lib().snmpc_eval(self._h, float(stop), _dp(x), _dp(u), _dp(xn), _dp(y), _dp(h))
        return xn, y, float(h[0])

    def solve_debug(self):
        N = self.N; nv, m = 2 * N, 3 * N
        buf = np.zeros(nv * nv + nv + m * nv + m)
        lib().snmpc_set_debug(_dp(buf))
        try:
            st = lib().snmpc_solve(self._h)
        finally:
            lib().snmpc_set_debug(None)
        o = 0
        H = buf[o:o + nv * nv].reshape(nv, nv); o += nv * nv
        q = buf[o:o + nv]; o += nv
        C = buf[o:o + m * nv].reshape(m, nv); o += m * nv
        d = buf[o:o + m]
        return st, dict(H=H, q=q, C=C, d=d)

    @property
    def cost(self):
        return float(self._view("cost")[0])

    @property
    def qp_iter(self):
        return lib().snmpc_qp_iter(self._h)

    @property
    def status(self):
        return lib().snmpc_status(self._h)


def global_work(reset=False):
    """(QPs, factorisations, back-solves) of every interior point solve of this process since the last reset (study counters)"""
    L = lib()
    out = (ctypes.c_long * 3)()
    L.oracle_global_work(out, int(bool(reset)))
    return tuple(out)
