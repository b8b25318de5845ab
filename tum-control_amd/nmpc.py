"""
nmpc.py -- host-side mirror of the reference's nominal NMPC controller interface, on top of the
HIP solver. Same names, argument meaning and return values as
  Model_Predictive_Controller/Nominal_NMPC/NMPC_STM_acados_settings.py:16,245   acados_settings(...)
  Model_Predictive_Controller/Nominal_NMPC/NMPC_class.py:34-317                 Nonlinear_Model_Predictive_Controller
so that main.py / Utils/SimulationMode_main_class.py drive it unchanged. (The RL weight-switching
branch, NMPC_class.py:120-160,208-239, is out of scope: it needs stable_baselines3 and is ML around
the controller, not the solve.)
"""
import types

import numpy as np

from . import config as _config
from .solver import ALL_STAGES, BatchedOcpSolver


class _Sym:
    """stand-in for the CasADi symbol vectors main.py only asks the size of (`MPC.model.u.size()[0]`, main.py:41)"""

    def __init__(self, n):
        self._n = int(n)

    def size(self):
        return (self._n, 1)

    @property
    def shape(self):
        return (self._n, 1)


def model_namespaces(cfg, nx=8, name="pred_dynamic_bicycle_model"):
    """(constraints, model) with the attributes the reference's harness reads from the controller
    (main.py:41, Utils/SimulationMode_main_class.py, Utils/Logging_Plotting.py)."""
    veh = cfg["veh"]
    constraints = types.SimpleNamespace(lat_acc_min=veh["lat_acc_min"], lat_acc_max=veh["lat_acc_max"],
                                        alat=lambda x: x[3] * x[5], a_lat=lambda x: x[3] * x[5])
    model = types.SimpleNamespace(
        name=name, nx=nx, nu=2, x=_Sym(nx), u=_Sym(2),
        jerk_min=veh["jerk_min"], jerk_max=veh["jerk_max"], acc_min=veh["acc_min"], acc_max=veh["acc_max"],
        delta_f_min=veh["delta_f_min"], delta_f_max=veh["delta_f_max"],
        delta_f_dot_min=veh["delta_f_dot_min"], delta_f_dot_max=veh["delta_f_dot_max"],
        params=types.SimpleNamespace(lf=veh["lf"], lr=veh["lr"], m=veh["m"], Iz=veh["Iz"],
                                     veh_length=veh["veh_length"], veh_width=veh["veh_width"], **cfg["tire"]),
        x0=np.zeros(nx))
    return constraints, model


def check_costfunction_type(mpc_params):
    """NMPC_class.py:90-94 picks the OCP by MPC_params['costfunction_type']: 'NONLINEAR_LS' (the shipped YAML,
    Config/EDGAR/MPC_params.yaml:1) builds NMPC_STM_acados_settings.py -- the formulation this library solves -- anything else
    the EXTERNAL-cost development variant (NMPC_STM_acados_settings_dev_lonlat.py:90-91, yref fed through set(j, "p", ...),
    NMPC_class.py:173-180), which is not built. Refuse it loudly rather than solve a different problem than configured."""
    t = mpc_params.get("costfunction_type", "NONLINEAR_LS")
    if t != "NONLINEAR_LS":
        raise NotImplementedError(f"costfunction_type '{t}': only 'NONLINEAR_LS' (NMPC_STM_acados_settings.py) is built; the EXTERNAL-cost "
                                  "variant (NMPC_STM_acados_settings_dev_lonlat.py) is out of scope")


def acados_settings(Tf, N, x0, Q, R, Qe, L1_pen, L2_pen, ax_max_interpolant=None, ay_max_interpolant=None,
                    combined_acc_limits=2, veh_params_file=None, tire_params_file=None,
                    solver_generate_C_code=True, solver_build=True, cfg=None, batch=1, device=0,
                    store_qp_in=False):
    """Builds the nominal OCP solver (NMPC_STM_acados_settings.py:16-245) and returns
    (constraints, model, acados_solver, ocp) like the reference. The interpolant / file / codegen
    arguments are accepted for signature compatibility; the gg table and vehicle constants come
    from `cfg` (config.default_config() or config.load_reference_config())."""
    if combined_acc_limits != 2:
        # 0 / 1 have two h rows per stage; in the reference itself they cannot construct a solver: the stage-0 slack
        # penalties are sized nh + 0 (NMPC_STM_acados_settings.py:192-198) while stage 0 has exactly one soft constraint
        # (ns_0 = 1), so the sizes only agree for nh = 1, the shipped variant 2
        raise NotImplementedError("only combined_acc_limits == 2 (circle), the shipped variant, is built")
    cfg = cfg or _config.default_config()
    veh = cfg["veh"]
    solver = BatchedOcpSolver(N=N, dt=Tf / N, nsub=3, batch=batch, device=device, cfg=cfg, store_qp_in=store_qp_in)
    solver.install_reference_ocp(Q=np.asarray(Q), R=np.asarray(R), Qe=np.asarray(Qe), L1=L1_pen, L2=L2_pen, w_scale=0.01)
    solver.constraints_set(0, "lbx", np.asarray(x0, dtype=float))
    solver.constraints_set(0, "ubx", np.asarray(x0, dtype=float))
    solver.cold_start()          # acados create: x_k = x0, u = 0
    constraints, model = model_namespaces(cfg)
    ocp = types.SimpleNamespace(cost=types.SimpleNamespace(cost_type="NONLINEAR_LS"), dims=types.SimpleNamespace(N=N),
                                nh=1, nh_e=1)
    return constraints, model, solver, ocp


CALL_PATTERNS = ("step", "acados")


def ref_rows(current_ref_traj, N):
    """the planner's dict (Utils/MPC_sim_utils.py:137-194: pos_x, pos_y, ref_yaw, ref_v) as (N+1, 6) yref rows [x, y, yaw, v, 0, 0]"""
    y = np.zeros((N + 1, 6))
    for col, key in enumerate(("pos_x", "pos_y", "ref_yaw", "ref_v")):
        y[:, col] = np.asarray(current_ref_traj[key][:N + 1])
    return y


def acados_call_sequence(ctl, current_ref_traj, stage_parameter=None):
    """One control step as the LITERAL AcadosOcpSolver call sequence of the reference's controller classes -- NMPC_class.py:169-206,
    SNMPC_class.py:181-214: N x set(j, "yref", 6 values) [+ set(j, "p", ...)], set(N, "yref", 4 values) [+ "p"], solve(),
    get(0, "u"), N x get(j, "x") when the solve succeeded, get_cost(), get_stats('time_tot' | 'sqp_iter' | 'qp_iter') -- every call a
    synchronous round trip through the C-ABI (2 N + 7 of them; 3 N + 8 with the parameter vector). What a maintainer gets who swaps
    ONLY the solver object and keeps the reference's controller class; `call_pattern="acados"` of the mirrored classes runs it, the
    tests hold it bit-equal to the one-call step (tum_ocp_step_async) and to the logged acados outputs.
    stage_parameter(j) -> the per-stage parameter vector of the SNMPC OCP (None: nominal OCP)."""
    s, N = ctl._solver, ctl.N
    y = ref_rows(current_ref_traj, N)
    for j in range(N + 1):
        s.set(j, "yref", y[j] if j < N else y[N, :4])
        if stage_parameter is not None:
            s.set(j, "p", stage_parameter(j))
    status = s.solve()
    u0 = s.get(0, "u")
    if status == 0:
        ctl.pred_X = np.vstack([np.asarray(s.get(j, "x"))[:ctl.nx].reshape(1, -1) for j in range(N)])
    ctl.stats[0] = s.get_cost()
    ctl.stats[1] = s.get_stats('time_tot')
    ctl.stats[2] = s.get_stats('sqp_iter')
    ctl.stats[3] = np.max(s.get_stats('qp_iter'))
    ctl.stats[4] = status
    return u0, ctl.pred_X, ctl.stats


def one_call_step(ctl, current_ref_traj):
    """The same control step as ONE enqueue and ONE wait (tum_ocp_step_async + tum_ocp_results_wait): the pending initial state, the
    reference, the solve and the read-back of u0 / predictions / cost / status, inputs and results through pinned memory the capsule
    owns -- instead of the 2 N + 7 synchronous round trips of acados_call_sequence."""
    s, N = ctl._solver, ctl.N
    summ, X, U = s.step(x0=ctl._x0_pending, yref=ref_rows(current_ref_traj, N), with_iterate=True)
    ctl._x0_pending = None
    status = int(np.max(summ[:, 3]))
    s.status = status
    u0 = np.array(U[0, 0])                                  # batch = 1
    if status == 0:
        ctl.pred_X = np.array(X[0, :N])
    ctl.stats[0] = float(summ[0, 2])
    ctl.stats[1] = s.get_stats('time_tot')
    ctl.stats[2] = s.get_stats('sqp_iter')
    ctl.stats[3] = float(np.max(summ[:, 4]))
    ctl.stats[4] = status
    return u0, ctl.pred_X, ctl.stats


class _SolverHandle:
    """`controller.acados_solver`: reading the attribute from OUTSIDE the class first flushes an initial state that still waits for the
    next one-call step (set_initial_state of call_pattern "step"), so code that follows the reference's other pattern --
    set_initial_state(x); controller.acados_solver.solve() -- or reads bounds / uploads device buffers sees the solver in the state the
    reference's eager setters would have left it in. The classes themselves use `_solver`."""

    @property
    def acados_solver(self):
        self._flush_x0()
        return self._solver

    @acados_solver.setter
    def acados_solver(self, s):
        self._solver = s


class Nonlinear_Model_Predictive_Controller(_SolverHandle):
    """NMPC_class.py:34-317. `sim_main_params` needs Tp, Ts, Ts_MPC (dict); config_path / MPC_params_file
    may point at a TUM-CONTROL Config directory, or be None to use the built-in EDGAR constants.
    call_pattern: "step" (default) -- a control step is ONE enqueue and ONE wait (tum_ocp_step_async); "acados" -- the reference's
    literal per-stage set / solve / get sequence (acados_call_sequence), eager setters; identical results."""

    def __init__(self, config_path=None, MPC_params_file=None, sim_main_params=None, X0_MPC=None, device=0,
                 store_qp_in=False, call_pattern="step"):
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
        self.MPC_params = self.cfg["mpc"]
        check_costfunction_type(self.MPC_params)
        self.Tp, self.Ts, self.Ts_MPC = sim["Tp"], sim["Ts"], sim["Ts_MPC"]
        self.N = int(self.Tp / self.Ts_MPC)
        m = self.MPC_params
        self.L1_pen, self.L2_pen = m["L1_pen"], m["L2_pen"]
        self.Q = np.diag([m["q_lon"] / m["s_lon"] ** 2, m["q_lat"] / m["s_lat"] ** 2,
                          m["q_yaw"] / m["s_yaw"] ** 2, m["q_vel"] / m["s_vel"] ** 2])
        self.R = np.diag([m["r_jerk"] / m["s_jerk"] ** 2, m["r_steering_rate"] / m["s_steering_rate"] ** 2])
        self.Qe = self.Q
        self.combined_acc_limits = m["combined_acc_limits"]
        self._device, self._store_qp_in = device, store_qp_in
        X0_MPC = np.zeros(8) if X0_MPC is None else np.asarray(X0_MPC, dtype=float)
        self.constraint, self.model, self.acados_solver, self.ocp = acados_settings(
            self.Tp, self.N, X0_MPC, self.Q, self.R, self.Qe, self.L1_pen, self.L2_pen,
            combined_acc_limits=self.combined_acc_limits, cfg=self.cfg, device=device, store_qp_in=store_qp_in)
        self.costfunction_type = self.ocp.cost.cost_type
        self.nx = 8
        self.x0 = X0_MPC
        self._solver.constraints_set(0, "lbx", self.x0)
        self._solver.constraints_set(0, "ubx", self.x0)
        self.stats = np.zeros(5)
        self.pred_X = np.empty((0, self.nx))
        self.nh, self.nh_e = 1, 1
        self.WMPC = False

    def solve(self, current_ref_traj):
        """NMPC_class.py:163-241: set yref on every stage, one SQP-RTI step, read u0 / predictions / stats."""
        if self.call_pattern == "acados":
            return acados_call_sequence(self, current_ref_traj)
        return one_call_step(self, current_ref_traj)

    def set_initial_state(self, x0):
        """NMPC_class.py:243-246 (lbx_0 = ubx_0 = x0). call_pattern "step": the state rides with the next solve() -- it goes up in the
        same enqueue as the reference trajectory (tum_ocp_step_async) instead of two synchronous setter calls of its own; reading
        `acados_solver` from outside, reset() and update_cost_function_weights() flush it first (_flush_x0). "acados": set at once."""
        self.x0 = x0
        self._x0_pending = np.array(x0, dtype=float).reshape(-1)
        if self.call_pattern == "acados":
            self._flush_x0()

    def _flush_x0(self):
        if getattr(self, "_x0_pending", None) is not None:
            x0, self._x0_pending = self._x0_pending, None
            self._solver.constraints_set(0, "lbx", x0)
            self._solver.constraints_set(0, "ubx", x0)

    def reset(self, x0):
        self._solver.reset()
        self.set_initial_state(x0)
        self._flush_x0()
        for i in range(self.N + 1):
            self._solver.set(i, 'x', self.x0)

    def reintialize_solver(self, X0_MPC, solver_generate_C_code=False, solver_build=False):
        self.constraint, self.model, self.acados_solver, self.ocp = acados_settings(
            self.Tp, self.N, X0_MPC, self.Q, self.R, self.Qe, self.L1_pen, self.L2_pen,
            combined_acc_limits=self.combined_acc_limits, cfg=self.cfg, device=self._device,
            store_qp_in=self._store_qp_in)
        self.set_initial_state(X0_MPC)

    def update_cost_function_weights(self, params):
        """NMPC_class.py:269-317: params = [q_xy, q_yaw, q_vel, r_jerk, r_steering_rate, L1, L2]; W is
        installed RAW (no 0.01 factor), slack penalties on every stage."""
        if hasattr(params, "numpy"):
            params = params.numpy()
        params = np.asarray(params, dtype=float)
        Q = np.diag([params[0], params[0], params[1], params[2]])
        R = np.diag([params[3], params[4]])
        L1, L2 = params[5], params[6]
        W = np.zeros((6, 6)); W[:4, :4] = Q; W[4:, 4:] = R
        s, N = self.acados_solver, self.N          # (the property: a pending initial state is flushed first)
        if self.call_pattern == "acados":          # NMPC_class.py:294-296: one cost_set per stage
            for i in range(N):
                s.cost_set(i, 'W', W)
        else:                                      # the same W on the stages 0..N-1 in ONE call (TUM_ALL_STAGES)
            s.cost_set(ALL_STAGES, 'W', W)
        s.cost_set(N, 'W', Q)
        z0, Z0 = np.ones(self.nh) * L1, np.ones(self.nh) * L2
        for f, v in (('zl', z0), ('zu', z0), ('Zl', Z0), ('Zu', Z0)):
            s.cost_set(0, f, v)
        ze, Ze = np.ones(self.nh_e + 1) * L1, np.ones(self.nh_e + 1) * L2
        for f, v in (('zl', ze), ('zu', ze), ('Zl', Ze), ('Zu', Ze)):
            s.cost_set(N, f, v)
        z, Z = np.ones(self.nh + 2) * L1, np.ones(self.nh + 2) * L2
        for i in range(1, N):
            for f, v in (('zl', z), ('zu', z), ('Zl', Z), ('Zu', Z)):
                s.cost_set(i, f, v)
