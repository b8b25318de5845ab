"""
closed_loop.py -- counterpart of the reference's closed-loop harness for a BATCH of independent vehicles
(SURVEY.md 8(f2), 8(f4)): planner -> batched SQP-RTI solve on the GPU -> plant step -> state estimation.

Restates, for the code around the hot path only what is needed to drive it the way the reference does:
  main.py:48-78 / Learning_To_Adapt/SafeRL_WMPC/get_baseline_performances.py:101-131   the loop
  Utils/SimulationMode_main_class.py:106-156                                            sim_step (simMode 0), StateEstimation
  Vehicle_Simulator/sim_model_dynamic_stm_pacejka.py:137-195 + VehicleSimulator.py:73-77 plant: 7-state single track
                                                                                         (input = acceleration, steering rate),
                                                                                         CasADi 'rk' with 4 finite elements = RK4 x 4
  Utils/Logging_Plotting.py:124-146,321-372                                             what gets logged, and the npz file (save)
  Utils/MPC_sim_utils.py:15-99                                                          disturbance set-up and generators
Every instance may carry its own cost weights / penalties (the BO / RL weight sweep as one batch).
"""
import numpy as np

from . import config as _config
from .planner import load_track, planner_emulator, yref_from_ref
from .solver import BatchedOcpSolver, CoupledSnmpcSolver, DeviceClosedLoop

WINDOWS = (1, 1, 4, 2, 2, 3, 4, 2)          # SimulationMode_main_class.py:86


def plant_xdot(x, a, sr, cfg):
    """xdot of the 7-state plant [posx,posy,yaw,vlong,vlat,yawrate,delta_f]; x: (B,7), a, sr: (B,)."""
    veh, tire, ph = cfg["veh"], cfg["tire"], cfg["phys"]
    lf, lr, m, Iz = veh["lf"], veh["lr"], veh["m"], veh["Iz"]
    yaw, vl, vt, r, de = x[:, 2], x[:, 3], x[:, 4], x[:, 5], x[:, 6]
    v = np.sqrt(vl ** 2 + vt ** 2) * 3.6
    fr = ph["fr0"] + ph["fr1"] * v / 100 + ph["fr4"] * (v / 100) ** 4
    Fz_f = m * lr * ph["g"] / (lf + lr); Fz_r = m * lf * ph["g"] / (lf + lr)
    Fx_f = -fr * Fz_f
    Fx_r = m * a - fr * Fz_r
    Faero = 0.5 * veh["ro"] * veh["S"] * veh["Cd"] * vl ** 2
    ok = vl > 0.001
    vls = np.where(ok, vl, 1.0)
    al_f = np.where(ok, de - np.arctan((vt + lf * r) / vls), 0.0)
    al_r = np.where(ok, np.arctan((lr * r - vt) / vls), 0.0)
    Bf, Cf, Df, Ef = tire["Bf"], tire["Cf"], tire["Df"], tire["Ef"]
    Br, Cr, Dr, Er = tire["Br"], tire["Cr"], tire["Dr"], tire["Er"]
    Fy_f_lat = Df * np.sin(Cf * np.arctan(Bf * al_f - Ef * (Bf * al_f - np.arctan(Bf * al_f))))
    Fy_r_lat = Dr * np.sin(Cr * np.arctan(Br * al_r - Er * (Br * al_r - np.arctan(Br * al_r))))
    Fmax_f = np.sqrt(Fz_f ** 2 + (Cf * Fz_f) ** 2); Fmax_r = np.sqrt(Fz_r ** 2 + (Cr * Fz_r) ** 2)
    Gy_f = np.clip(Fx_f / Fmax_f, -0.98, 0.98); Gy_r = np.clip(Fx_r / Fmax_r, -0.98, 0.98)
    Fy_f = Fy_f_lat * np.cos(np.arcsin(Gy_f)); Fy_r = Fy_r_lat * np.cos(np.arcsin(Gy_r))
    xd = np.empty_like(x)
    xd[:, 0] = vl * np.cos(yaw) - vt * np.sin(yaw)
    xd[:, 1] = vl * np.sin(yaw) + vt * np.cos(yaw)
    xd[:, 2] = r
    xd[:, 3] = (Fx_r - Faero - Fy_f * np.sin(de) + Fx_f * np.cos(de) + m * vt * r) / m
    xd[:, 4] = (Fy_r + Fy_f * np.cos(de) + Fx_f * np.sin(de) - m * vl * r) / m
    xd[:, 5] = (lf * (Fy_f * np.cos(de) + Fx_f * np.sin(de)) - lr * Fy_r) / Iz
    xd[:, 6] = sr
    return xd


def plant_step(x, a, sr, cfg, Ts=0.02, n_elem=4, w=None):
    """One simulator step: classic RK4 with `n_elem` equal sub-steps over Ts, inputs held constant. w (B,7): additive disturbance
    of the state derivatives, constant over the step (simulator_step_disturbed: xdot + w, sim_model_dynamic_stm_pacejka.py:196)."""
    h = Ts / n_elem
    x = x.copy()
    f = (lambda y: plant_xdot(y, a, sr, cfg)) if w is None else (lambda y: plant_xdot(y, a, sr, cfg) + w)
    for _ in range(n_elem):
        k1 = f(x)
        k2 = f(x + 0.5 * h * k1)
        k3 = f(x + 0.5 * h * k2)
        k4 = f(x + h * k3)
        x = x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


STATE_NAMES = ("posx", "posy", "yaw", "vlong", "vlat", "yawrate", "delta_f")


def sample_from_ellipsoid(w, Z, rng):
    """Utils/MPC_sim_utils.py:70-86: a point of the ellipsoid with centre w and shape matrix Z -- radius rand()^(1/n), direction a
    normalised Gaussian vector, mapped through the eigen-decomposition of Z. Same order of random draws as the reference (one rand(),
    then n randn())."""
    n = w.shape[0]
    lam, v = np.linalg.eig(Z)
    r = rng.rand() ** (1 / n)
    x = rng.randn(n)
    x = x / np.linalg.norm(x)
    x *= r
    return v @ (np.sqrt(lam) * x) + w


def generate_disturbances(bounds, kind, rng):
    """Utils/MPC_sim_utils.py:54-67: one realisation for the n channels of `bounds` ([-b, b] pairs). 'uniform' draws from the ELLIPSOID
    with semi-axes sqrt(b) (the reference passes diag(b) as the shape matrix), 'gaussian' has standard deviation b, 'absolute'
    returns b; anything else is uniform in the box."""
    b = np.asarray(bounds, dtype=float)
    if kind == 'uniform':
        return sample_from_ellipsoid(np.zeros(len(b)), np.diag(b.T[1, :]), rng)
    if kind == 'gaussian':
        return np.array([rng.normal(0, b[j][1]) for j in range(len(b))])
    if kind == 'absolute':
        return np.array([b[j][1] for j in range(len(b))])
    return np.array([rng.uniform(b[j][0], b[j][1]) for j in range(len(b))])


class DisturbanceModel:
    """The disturbance set-up of the reference's harness (initDisturbanceSim, Utils/MPC_sim_utils.py:15-51; switches and magnitudes:
    Config/EDGAR/sim_main_params.yaml:44-80) and the realisations sim_step draws from it (SimulationMode_main_class.py:121-143).
    `draw` pre-draws a whole run: the device loop plays a realisation back (as the reference does with disturbance_playback),
    the host loop applies the same arrays step by step."""

    def __init__(self, sim_main_params=None, **override):
        p = dict(_config.SIM)
        p.update(sim_main_params or {})
        p.update(override)
        self.derivatives = bool(p["simulate_disturbances"])
        self.state_estimation = bool(p["simulate_state_estimation"])
        self.types = [p["disturbance_type_derivatives"], p["disturbance_type_state_estimation"]]
        self.bounds_derivatives = [[-p[f"w_{n}_dot"], p[f"w_{n}_dot"]] for n in STATE_NAMES]
        self.bounds_state_estimation = [[-p[f"w_{n}"], p[f"w_{n}"]] for n in STATE_NAMES]

    def draw(self, n_steps, batch=1, seed=0):
        """(w_deriv, e_est): (n_steps, batch, 7) arrays or None. Vehicle b uses numpy's legacy generator seeded with seed + b and
        draws in the reference's order -- per control step the derivative disturbance first, then the estimation error -- so
        batch = 1 reproduces what main.py produces after np.random.seed(seed)."""
        w = np.zeros((n_steps, batch, 7)) if self.derivatives else None
        e = np.zeros((n_steps, batch, 7)) if self.state_estimation else None
        for b in range(batch):
            rng = np.random.RandomState(seed + b)
            for i in range(n_steps):
                if w is not None:
                    w[i, b] = generate_disturbances(self.bounds_derivatives, self.types[0], rng)
                if e is not None:
                    e[i, b] = generate_disturbances(self.bounds_state_estimation, self.types[1], rng)
        return w, e


def lon_lat_deviations(ego_yaw, ego_x, ego_y, ref_x, ref_y):
    """Utils/MPC_sim_utils.py:103-112: the deviation vector rotated into the vehicle frame"""
    c, s_ = np.cos(-ego_yaw), np.sin(-ego_yaw)
    return c * (ref_x - ego_x) - s_ * (ref_y - ego_y), s_ * (ref_x - ego_x) + c * (ref_y - ego_y)


def wrap_yaw(yaw):
    """postprocess_yaw (Utils/MPC_sim_utils.py:124-134): fmod into (-2 pi, 2 pi), negatives shifted up"""
    y = np.fmod(np.asarray(yaw, dtype=float), 2 * np.pi)
    return np.where(y < 0, y + 2 * np.pi, y)


LOG_KEYS = ("MPC_SimX", "CiLX", "simU", "simREF", "simSolverDebug", "sim_disturbance_derivatives", "sim_disturbance_state_estimation",
            "a_lat", "dev_lat", "dev_long", "dev_vel", "dev_yaw", "t")


def log_file_arrays(logs, b, w_deriv=None, e_est=None, T=None, Ts=0.02, drop_last_step=True):
    """The arrays of one vehicle's `full_logs.npz` as Logger.save_logs writes them (Utils/Logging_Plotting.py:321-395): the five raw
    logs of instance b (yaw columns wrapped to [0, 2 pi) as the reference stores them), the disturbance realisation, and the derived
    channels a_lat = vlong * yawrate, dev_lat / dev_long (vehicle frame), dev_vel, dev_yaw, t.
    drop_last_step: Logger.truncate cuts at `current_step`, the INDEX of the last control step -- a run of n steps is stored as n - 1
    rows of simU / simREF / simSolverDebug and n rows of CiLX / MPC_SimX (the reference's own files: 5500 steps run, 5499 stored), while
    the disturbance arrays keep all n rows and t = linspace(0, T, n - 1). True reproduces that layout; False keeps every step."""
    simU = np.array(logs["simU"][:, b])
    n_run = len(simU)
    n = n_run - 1 if (drop_last_step and n_run > 1) else n_run
    CiLX = np.array(logs["CiLX"][:n + 1, b]); SimX = np.array(logs["MPC_SimX"][:n + 1, b])
    simU = simU[:n]; simREF = np.array(logs["simREF"][:n, b]); dbg = np.array(logs["simSolverDebug"][:n, b])
    dev_vel = np.abs(CiLX[1:, 3] - simREF[:, 3])
    CiLX[:, 2] = wrap_yaw(CiLX[:, 2]); SimX[:, 2] = wrap_yaw(SimX[:, 2])
    dev_yaw = np.abs(CiLX[1:, 2] - simREF[:, 2])
    dev_long, dev_lat = lon_lat_deviations(CiLX[1:, 2], CiLX[1:, 0], CiLX[1:, 1], simREF[:, 0], simREF[:, 1])

    def realisation(a):
        out = np.zeros((n_run, 7))
        if a is not None:
            a = np.asarray(a)[:n_run, b]
            out[:len(a)] = a
        return out
    return dict(MPC_SimX=SimX, CiLX=CiLX, simU=simU, simREF=simREF, simSolverDebug=dbg,
                sim_disturbance_derivatives=realisation(w_deriv), sim_disturbance_state_estimation=realisation(e_est),
                a_lat=CiLX[:, 3] * CiLX[:, 5], dev_lat=dev_lat, dev_long=dev_long, dev_vel=dev_vel, dev_yaw=dev_yaw,
                t=np.linspace(0.0, _config.SIM["T"] if T is None else T, n))      # (Logging_Plotting.py:348-349: always sim_main_params['T'], whatever the run length)


class MovingAverageEstimator:
    """StateEstimation (SimulationMode_main_class.py:152-156): per-state moving average over the last
    WINDOWS[i] samples (fewer while the buffer fills), buffers start empty."""

    def __init__(self, batch):
        self.hist = [[] for _ in range(8)]
        self.batch = batch

    def __call__(self, x_next):
        out = np.empty_like(x_next)
        for i in range(8):
            self.hist[i].append(x_next[:, i].copy())
            if len(self.hist[i]) > 15:
                self.hist[i].pop(0)
            out[:, i] = np.mean(self.hist[i][-WINDOWS[i]:], axis=0)
        return out


class ClosedLoopBatch:
    """B independent closed loops on one track, one OCP instance each; `params` (B,7) are per-instance
    [q_xy, q_yaw, q_vel, r_jerk, r_steer, L1, L2] as in update_cost_function_weights (None: YAML defaults x0.01)."""

    def __init__(self, track_name, batch=1, params=None, N=38, Tp=3.04, Ts=0.02, idx_start=0, cfg=None, device=0,
                 on_device=False, log_capacity=0, controller="nominal", disturbances=None, disturbance_steps=0, seed=0,
                 qp_warm_start=None, qp_tol=None):
        self.cfg = cfg or _config.default_config()
        kw = {} if qp_tol is None else dict(qp_tol=tuple(qp_tol))          # (termination tolerances of the interior point method; default: the solver's 1e-8)
        self.track = load_track(track_name)
        self.B, self.N, self.Tp, self.Ts = batch, N, Tp, Ts
        tr = self.track
        x0 = np.array([tr[idx_start, 0], tr[idx_start, 1], np.mod(tr[idx_start, 2], 2 * np.pi), tr[idx_start, 3], 0, 0, 0, 0.0])
        self.x_mpc = np.tile(x0, (batch, 1))                     # X0_MPC
        self.x_sim = self.x_mpc[:, :7].copy()                    # X0_sim
        self.pose = self.x_mpc[:, :2].copy()
        self.controller = controller
        if controller == "nominal":       # main.py:33-36 picks the controller class by MPC_params['MPC_type']
            self.solver = BatchedOcpSolver(N=N, dt=Tp / N, nsub=3, batch=batch, device=device, cfg=self.cfg, qp_warm_start=qp_warm_start, **kw)
        elif controller == "snmpc":       # the coupled SNMPC OCP; x0_samples = compute_x0dist(x0) before every solve
            from . import snmpc as _snm
            m = self.cfg["mpc"]
            stds = np.asarray(m["stds"], dtype=float)
            nvar = int(np.count_nonzero(stds))
            w = _snm.hammersley_normal(m["n_samples"], nvar)
            A = _snm.pce_matrix(w, _snm.alpha_generation(nvar, m["expansion_degree"]))
            self._x0_offsets = _snm.x0_offsets(w, stds)
            self.solver = CoupledSnmpcSolver(N=N, dt=Tp / N, batch=batch, Apce=A, uph=min(int(m["uncertainty_propagation_horizon"]), N),
                                             gamma=m["gamma"], device=device, cfg=self.cfg, x0_offsets=self._x0_offsets, qp_warm_start=qp_warm_start, **kw)
        elif controller == "r2":          # nominal OCP + covariance back-off after every solve (K7 attached to the solve)
            from .r2nmpc import r2_setup
            m, veh = self.cfg["mpc"], self.cfg["veh"]
            self.solver = BatchedOcpSolver(N=N, dt=Tp / N, nsub=3, batch=batch, device=device, cfg=self.cfg, store_qp_in=True,
                                           qp_warm_start=qp_warm_start, **kw)
            S0, BWB = r2_setup(m["stds"], Tp / N)
            self.solver.r2_attach(S0, BWB, int(m["uncertainty_propagation_horizon"]), veh["delta_f_min"], veh["delta_f_max"], 1.0)
        else:
            raise ValueError("controller must be 'nominal', 'snmpc' or 'r2'")
        self.solver.install_reference_ocp()
        if params is not None:
            self.set_weights(np.asarray(params, dtype=float).reshape(batch, 7))
        self.solver.set_x0(self.x_mpc)
        self.solver.cold_start()
        self.est = MovingAverageEstimator(batch)
        # on_device: planner, plant and estimator run as kernels next to the solve (no host round trip per step)
        self.dev = None
        if on_device:
            self.dev = DeviceClosedLoop(self.solver, self.track, Tp, Ts=Ts, n_elem=4, windows=WINDOWS, log_capacity=log_capacity)
            self.dev.set_state(self.x_sim, self.x_mpc, cold_start=True)
        self.log = dict(CiLX=[self.x_sim.copy()], MPC_SimX=[self.x_mpc.copy()], simU=[], simREF=[], simSolverDebug=[])
        # disturbance realisation (sim_step's simulate_disturbances / simulate_state_estimation): a DisturbanceModel -- `disturbance_steps`
        # control steps are drawn with `seed` -- or a (w_deriv, e_est) pair of (n_steps, B, 7) arrays / None to play back
        self.w_deriv = self.e_est = None
        self._i = 0
        self._last_logs = None
        if disturbances is not None:
            w, e = disturbances.draw(disturbance_steps, batch, seed) if isinstance(disturbances, DisturbanceModel) else disturbances
            self.set_disturbances(w, e)

    def set_disturbances(self, w_deriv=None, e_est=None):
        self.w_deriv = None if w_deriv is None else np.ascontiguousarray(w_deriv, dtype=float).reshape(-1, self.B, 7)
        self.e_est = None if e_est is None else np.ascontiguousarray(e_est, dtype=float).reshape(-1, self.B, 7)
        if self.dev is not None:
            self.dev.set_disturbances(self.w_deriv, self.e_est)

    def set_weights(self, p):
        s, B, N = self.solver, self.B, self.N
        W = np.zeros((B, 6, 6))
        for j in range(B):
            W[j] = np.diag([p[j, 0], p[j, 0], p[j, 1], p[j, 2], p[j, 3], p[j, 4]])
        s.cost_set(-1, "W", W if B > 1 else W[0])          # (-1 = ALL_STAGES: the stages 0..N-1)
        s.cost_set(N, "W", W[:, :4, :4] if B > 1 else W[0, :4, :4])
        for st, n in ((0, 1), (1, 3), (N, 2)):
            for f, col in (("zl", 5), ("zu", 5), ("Zl", 6), ("Zu", 6)):
                v = np.repeat(p[:, col:col + 1], n, axis=1)
                s.cost_set(st, f, v if B > 1 else v[0])

    def step(self):
        s, B, N = self.solver, self.B, self.N
        yref = np.zeros((B, N + 1, 6)); ref0 = np.zeros((B, 4))
        for b in range(B):
            _, ref = planner_emulator(self.track, self.pose[b], N + 1, self.Tp, True)
            yref[b] = yref_from_ref(ref, N); ref0[b] = ref[0]
        s.set_yref_all(yref if B > 1 else yref[0])
        status = s.solve()
        X, U = s.get_iterate()
        u0, x1 = U[:, 0], X[:, 1]
        stats = np.stack([np.atleast_1d(s.get_cost()), np.full(B, s.get_stats("time_tot")), np.ones(B),
                          s.get_stats("qp_iter").astype(float), s.get_stats("status").astype(float)], axis=1)
        # sim_step, simMode 0: the plant takes the predicted acceleration of stage 1 and the steering rate
        a_in, sr_in = x1[:, 7].copy(), u0[:, 1].copy()
        failed = np.nonzero(stats[:, 4] != 0)[0]
        if len(failed):
            self._reinitialise(failed, X, U)
        x_sim_next = plant_step(self.x_sim, a_in, sr_in, self.cfg, self.Ts)
        # SimulationMode_main_class.py:121-143: the TRUE state follows the undisturbed step; what the estimator is fed is a second step
        # with disturbed derivatives (if simulated) plus the state estimation error (if simulated)
        i = self._i
        x_meas = x_sim_next
        if self.w_deriv is not None and i < len(self.w_deriv):
            x_meas = plant_step(self.x_sim, a_in, sr_in, self.cfg, self.Ts, w=self.w_deriv[i])
        if self.e_est is not None and i < len(self.e_est):
            x_meas = x_meas + self.e_est[i]
        self._i += 1
        x_next = np.concatenate([x_meas, a_in[:, None]], axis=1)
        self.pose = x_sim_next[:, :2].copy()
        self.x_sim = x_sim_next
        self.x_mpc = self.est(x_next)
        s.set_x0(self.x_mpc if B > 1 else self.x_mpc[0])
        lg = self.log
        lg["simU"].append(u0.copy()); lg["simREF"].append(ref0); lg["simSolverDebug"].append(stats)
        lg["CiLX"].append(x_sim_next.copy()); lg["MPC_SimX"].append(x1.copy())
        return status

    def _reinitialise(self, failed, X, U):
        """main.py:59-61: `if MPC_stats[-1] != 0: MPC.reintialize_solver(x_next)` -- the failed instances get a fresh solver,
        cold-started at the state the failed solve started from (NMPC_class.py:256-267; sample copies included for the SNMPC
        controller, SNMPC_class.py:274-281; nominal bounds again for R2NMPC). The control of this step is still the failed
        solver's u0 and the last good prediction. Host mirror of what plant_advance_kernel does on the device."""
        s, N, B = self.solver, self.N, self.B
        X = X.copy(); U = U.copy()
        X[failed] = self.x_mpc[failed][:, None, :]
        U[failed] = 0.0
        s.set_iterate(X if B > 1 else X[0], U if B > 1 else U[0])
        if self.controller == "snmpc":
            ns = s.ns
            x0s = self.x_mpc[failed][:, None, :] + np.concatenate([np.zeros((1, 8)), self._x0_offsets])[None]    # (F, ns+1, 8)
            for k in range(N + 1):
                xk = np.asarray(s.get(k, "x")).reshape(B, 8 * (ns + 1))
                xk[failed] = x0s.reshape(len(failed), -1)
                s.set(k, "x", xk if B > 1 else xk[0])
        if self.controller == "r2":
            veh = self.cfg["veh"]
            for k in range(1, N):
                for f, val in (("lbx", veh["delta_f_min"]), ("ubx", veh["delta_f_max"]), ("uh", 1.0)):
                    v = np.atleast_1d(s.constraints_get(k, f)).copy()
                    v[failed] = val
                    s.constraints_set(k, f, v if B > 1 else v[:1])

    def run(self, n_steps):
        if self.dev is not None:
            self.dev.run(n_steps)
            self.x_sim, self.x_mpc, self.pose = self.dev.get("x_sim"), self.dev.get("x_mpc"), self.dev.get("pose")
            self._last_logs = self.dev.logs() if self.dev.log_capacity else None
            return self._last_logs
        for _ in range(n_steps):
            self.step()
        self._last_logs = {k: np.array(v) for k, v in self.log.items()}          # arrays are (steps[+1], B, dim)
        return self._last_logs

    def save(self, path, instance=None, T=None, drop_last_step=True):
        """Write the loop's logs as the reference's Logger.save_logs does (Utils/Logging_Plotting.py:321-372: np.savez with the keys
        LOG_KEYS, yaw wrapped, derived deviation channels). instance = b: one file `path` for vehicle b -- what Papers_Plots and the
        RL / BO tooling read; instance = None: one file per vehicle, `path` formatted with the vehicle index ('logs/{}.npz' ->
        logs/0.npz, ... -- the layout of Learning_To_Adapt/SafeRL_WMPC/_baseline/F/<track>/<k>.npz). drop_last_step: as Logger.truncate
        does (log_file_arrays). Returns the paths written."""
        logs = self._last_logs
        if logs is None:
            raise Exception("ClosedLoopBatch.save: no logs (run() first; a device loop needs log_capacity > 0)")
        out = []
        for b in (range(self.B) if instance is None else [int(instance)]):
            f = path.format(b) if instance is None else path
            np.savez(f, **log_file_arrays(logs, b, self.w_deriv, self.e_est, T=T, Ts=self.Ts, drop_last_step=drop_last_step))
            out.append(f)
        return out
