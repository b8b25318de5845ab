"""
snmpc.py -- scenario (sigma-point / Monte-Carlo) fan-out around the batched solver.

Host-side restatement of the one-off PCE set-up of the reference's stochastic controller
(Model_Predictive_Controller/Stochastic_NMPC/stochastic_mpc_utils.py:17-91, run once at construction,
SNMPC_class.py:39-177) plus the batch restatement of its per-step work: the sigma points become the
batch axis (independent 8-state OCPs sharing yref), the PCE mean / variance become a reduction over
each scenario group (SNMPC_acados_settings.py:116-133). chaospy (Hammersley sampling) is not available;
the sampling recipe below reproduces the sigma points stored in the reference's acados_ocp_SNMPC.json
(tests/golden/pce.npz) to 1e-14.
"""
import itertools
import math

import numpy as np
from scipy.special import ndtri

from . import config as _config
from .nmpc import CALL_PATTERNS, _SolverHandle, acados_call_sequence, one_call_step
from .solver import BatchedOcpSolver, CoupledSnmpcSolver


def hermite(x, n):
    """stochastic_mpc_utils.py:17-25 (note: divides by sqrt(n!) at EVERY recursion level, reproduced literally)."""
    if n == 0:
        return (1.0 + 0.0 * x) / math.sqrt(math.factorial(0))
    if n == 1:
        return x / math.sqrt(math.factorial(1))
    return (x * hermite(x, n - 1) - (n - 1) * hermite(x, n - 2)) / math.sqrt(math.factorial(n))


def alpha_generation(n_rand, degree):
    """Multi-indices with |alpha| <= degree, ascending total degree, ties in itertools.product order (:27-38)."""
    al = [a for a in itertools.product(range(degree + 1), repeat=n_rand) if sum(a) <= degree]
    al.sort(key=sum)                       # stable
    return np.array(al, dtype=int)


def pce_basis(w, alphas):
    """Phi(w) for one sample w (n_rand,) -> (L,) (:40-54, gaussian)."""
    return np.array([np.prod([hermite(w[j], int(a[j])) for j in range(len(a))]) for a in alphas])


def _vdc(i, base):
    v, f = 0.0, 1.0 / base
    while i > 0:
        v += (i % base) * f
        i //= base
        f /= base
    return v


_PRIMES = (2, 3, 5, 7, 11, 13, 17, 19)


def hammersley_normal(n_samples, n_rand):
    """chaospy.J(Normal(0,1)^n).sample(n, rule='hammersley') (:59-63): van-der-Corput in the first n-1 prime
    bases (burn-in = largest base used), last dimension equispaced, mapped through the normal inverse CDF.
    Returns (n_rand, n_samples)."""
    bases = _PRIMES[:max(n_rand - 1, 0)]
    burn = max(bases) if bases else 0
    u = np.zeros((n_rand, n_samples))
    for k in range(n_samples):
        for d, bse in enumerate(bases):
            u[d, k] = _vdc(burn + 1 + k, bse)
        u[n_rand - 1, k] = (k + 1) / (n_samples + 1)
    return ndtri(u)


def pce_matrix(w_samples, alphas):
    """A = inv(Phi' Phi) Phi' (L x n_samples), (:66-74)."""
    Phi = np.array([pce_basis(w_samples[:, i], alphas) for i in range(w_samples.shape[1])])
    return np.linalg.inv(Phi.T @ Phi) @ Phi.T


def x0_offsets(w_samples, stds):
    """(n_samples, 8) offsets stds (.) w on the states with non-zero std (compute_x0dist, :78-91)."""
    stds = np.asarray(stds, dtype=float)
    act = np.nonzero(stds)[0]
    off = np.zeros((w_samples.shape[1], 8))
    for s in range(w_samples.shape[1]):
        off[s, act] = stds[act] * w_samples[:, s]
    return off


def compute_x0dist(x0, w_samples, stds):
    """(n_samples+1, 8): row 0 = x0, row s = x0 + offsets[s-1]."""
    return np.vstack([np.asarray(x0, float)[None], np.asarray(x0, float)[None] + x0_offsets(w_samples, stds)])


class ScenarioSNMPC:
    """Batch restatement of the SNMPC step: P poses x (1 nominal + S scenarios) independent nominal OCPs."""

    def __init__(self, n_poses, n_samples=None, stds=None, degree=None, N=40, dt=0.08, nsub=3, device=0,
                 w_samples=None, cfg=None):
        cfg = cfg or _config.default_config()
        m = cfg["mpc"]
        self.stds = np.asarray(m["stds"] if stds is None else stds, dtype=float)
        self.n_rand = int(np.count_nonzero(self.stds))
        self.degree = m["expansion_degree"] if degree is None else degree
        self.alphas = alpha_generation(self.n_rand, self.degree)
        if w_samples is None:
            n_samples = m["n_samples"] if n_samples is None else n_samples
            w_samples = hammersley_normal(n_samples, self.n_rand)
        self.w = np.asarray(w_samples, dtype=float)
        self.S = self.w.shape[1]
        self.P = int(n_poses)
        self.A = pce_matrix(self.w, self.alphas) if self.S >= len(self.alphas) else None
        self.offsets = x0_offsets(self.w, self.stds)
        self.gamma = m["gamma"]
        self.kappa = math.sqrt((1 - self.gamma) / self.gamma)
        self.solver = BatchedOcpSolver(N=N, dt=dt, nsub=nsub, batch=self.P * (self.S + 1), device=device, cfg=cfg)
        self.solver.install_reference_ocp()
        self.N = N

    def solve(self, x0_poses, yref_poses, cold=True):
        """x0_poses (P,8), yref_poses (P,N+1,6). Returns status, nominal u0 (P,2), PCE mean / var of x_1 (P,8)."""
        s = self.solver
        s.set_x0_fanout(x0_poses, self.offsets)
        s.set_yref_all(np.repeat(np.asarray(yref_poses, float), self.S + 1, axis=0))
        if cold:
            s.cold_start()
        st = s.solve()
        X, U = s.get_iterate()
        mean = var = None
        if self.A is not None:
            mean, var = s.pce_moments("x", 1, self.A)
        return st, U[::self.S + 1, 0], mean, var


class Stochastic_Nonlinear_Model_Predictive_Controller(_SolverHandle):
    """Host-side mirror of the reference's SNMPC controller on the coupled OCP:
    Model_Predictive_Controller/Stochastic_NMPC/SNMPC_class.py:38-349, same method names, arguments and returns
    (`solve` -> u0, pred_X of the nominal copy, stats = [cost, time_tot, sqp_iter, max qp_iter, status]).
    The RL weight-switching branch (SNMPC_class.py:134-177,215-246) is out of scope as in nmpc.py.
    call_pattern as in nmpc.py: "step" (default, one enqueue and one wait per control step) or "acados" (the reference's literal
    sequence, SNMPC_class.py:181-214: set(j, "yref") AND set(j, "p") per stage, stacked 88-value lbx_0 / ubx_0 set at once)."""

    def __init__(self, config_path=None, MPC_params_file=None, sim_main_params=None, X0_MPC=None, device=0, call_pattern="step"):
        if call_pattern not in CALL_PATTERNS:
            raise ValueError(f"call_pattern must be one of {CALL_PATTERNS}")
        self.call_pattern = call_pattern
        self._x0_pending = None
        if config_path is not None and MPC_params_file is not None:
            self.cfg = _config.load_reference_config(config_path, sim_main_params, MPC_params_file)
        else:
            self.cfg = _config.default_config()
        sim = dict(self.cfg["sim"])
        if sim_main_params:
            sim.update({k: sim_main_params[k] for k in ("Tp", "Ts", "Ts_MPC") if k in sim_main_params})
        m = self.MPC_params = self.cfg["mpc"]
        from .nmpc import check_costfunction_type
        check_costfunction_type(m)
        self.Tp, self.Ts, self.Ts_MPC = sim["Tp"], sim["Ts"], sim["Ts_MPC"]
        self.N = int(self.Tp / self.Ts_MPC)
        self.L1_pen, self.L2_pen = m["L1_pen"], m["L2_pen"]
        self.Q = np.diag([m["q_lon"] / m["s_lon"] ** 2, m["q_lat"] / m["s_lat"] ** 2,
                          m["q_yaw"] / m["s_yaw"] ** 2, m["q_vel"] / m["s_vel"] ** 2])
        self.R = np.diag([m["r_jerk"] / m["s_jerk"] ** 2, m["r_steering_rate"] / m["s_steering_rate"] ** 2])
        self.Qe = self.Q
        if m["combined_acc_limits"] != 2:
            raise NotImplementedError("only combined_acc_limits == 2 (circle), the shipped variant, is built")
        # PCE set-up, SNMPC_class.py:81-104
        self.n_samples = m["n_samples"]
        self.stds = np.asarray(m["stds"], dtype=float)
        self.expansion_degree = m["expansion_degree"]
        self.n_vars = int(np.count_nonzero(self.stds))
        self.alphas = alpha_generation(self.n_vars, self.expansion_degree)
        self.num_poly_terms = len(self.alphas)
        self.w_samples = hammersley_normal(self.n_samples, self.n_vars)
        self.A = pce_matrix(self.w_samples, self.alphas)
        uph = int(m["uncertainty_propagation_horizon"])
        self.stop_flags = np.zeros(self.N + 1)
        self.stop_flags[uph:] = 1
        self.risk_parameter = np.array(m["gamma"]).reshape(1)
        X0_MPC = np.zeros(8) if X0_MPC is None else np.asarray(X0_MPC, dtype=float)
        self._uph = min(uph, self.N)
        self._device = device
        from .nmpc import model_namespaces
        self.constraint, self.model = model_namespaces(self.cfg, nx=8 * (self.n_samples + 1), name="SNMPC")
        self.nx = int(self.model.x.size()[0] / (self.n_samples + 1))          # SNMPC_class.py:113
        self.x0 = X0_MPC
        self.costfunction_type = "NONLINEAR_LS"
        self.nh, self.nh_e = 1, 1
        self.acados_solver = self._build_solver(X0_MPC)
        self.stats = np.zeros(5)
        self.pred_X = np.empty((0, self.nx))
        self.WMPC = False

    def _build_solver(self, X0_MPC):
        """acados_settings(self.Tp, self.N, x0_samples.flatten(), self.Q, self.R, self.Qe, self.L1_pen, self.L2_pen, ...)
        (SNMPC_class.py:106-111 at construction and :274-281 in reintialize_solver): a fresh solver built from THIS
        controller's horizon, weights, penalties and PCE set-up, cold-started at the sample states of X0_MPC."""
        m = self.MPC_params
        x0_samples = compute_x0dist(np.asarray(X0_MPC, dtype=float), self.w_samples, self.stds)
        s = CoupledSnmpcSolver(N=self.N, dt=self.Tp / self.N, batch=1, Apce=self.A, uph=self._uph,
                               gamma=m["gamma"], device=self._device, cfg=self.cfg)
        s.install_reference_ocp(Q=self.Q, R=self.R, Qe=self.Qe, L1=self.L1_pen, L2=self.L2_pen, w_scale=0.01)
        s.constraints_set(0, "lbx", x0_samples.flatten())
        s.constraints_set(0, "ubx", x0_samples.flatten())
        for i in range(self.N + 1):
            s.set(i, "p", self._stage_parameter(i))
        s.cold_start()                  # SNMPC_class.py:126-127: x_j = x0_samples for all j
        s.set_x0_offsets(x0_offsets(self.w_samples, self.stds))       # for the 8-value x0 of a step (set_initial_state)
        return s

    def _stage_parameter(self, j):
        """p_j = [A_pce.flatten(), risk_parameter, stop_flag_j] (SNMPC_class.py:124,185,193)"""
        return np.concatenate((self.A.flatten(), self.risk_parameter, self.stop_flags[j].reshape(1)))

    def solve(self, current_ref_traj):
        """SNMPC_class.py:179-257"""
        if self.call_pattern == "acados":
            return acados_call_sequence(self, current_ref_traj, stage_parameter=self._stage_parameter)
        return one_call_step(self, current_ref_traj)

    def set_initial_state(self, x0):
        """SNMPC_class.py:259-264 (lbx_0 = ubx_0 = compute_x0dist(x0)). The estimated state rides with the next solve()
        (tum_ocp_step_async) and is fanned out to the sample initial conditions on the device -- x0 + stds (.) w_s, the same
        sum compute_x0dist forms (tum_ocp_snmpc_set_offsets, set once in _build_solver); whoever touches the solver in between
        flushes it (_flush_x0)."""
        self.x0 = x0
        self._x0_pending = np.array(x0, dtype=float).reshape(-1)
        if self.call_pattern == "acados":
            self._flush_x0()

    def _flush_x0(self):
        if getattr(self, "_x0_pending", None) is not None:
            x0, self._x0_pending = self._x0_pending, None
            x0_samples = compute_x0dist(x0, self.w_samples, self.stds)
            self._solver.constraints_set(0, "lbx", x0_samples.flatten())
            self._solver.constraints_set(0, "ubx", x0_samples.flatten())

    def reset(self, x0):
        self._x0_pending = None
        self._solver.reset()
        x0_samples = compute_x0dist(x0, self.w_samples, self.stds)
        self._solver.constraints_set(0, "lbx", x0_samples.flatten())
        self._solver.constraints_set(0, "ubx", x0_samples.flatten())
        for i in range(self.N + 1):
            self._solver.set(i, 'x', x0_samples.flatten())

    def reintialize_solver(self, X0_MPC, solver_generate_C_code=False, solver_build=False):
        """SNMPC_class.py:274-281: a fresh solver with the SAME Q / R / N / penalties / PCE set-up, cold-started at the
        sample states of X0_MPC (called by main.py:59-61 after every failed solve)."""
        self.acados_solver = self._build_solver(X0_MPC)
        self.set_initial_state(X0_MPC)

    def update_cost_function_weights(self, params):
        """SNMPC_class.py:283-331 (same protocol as NMPC_class.py:269-317)"""
        from .nmpc import Nonlinear_Model_Predictive_Controller as _N
        _N.update_cost_function_weights(self, params)
