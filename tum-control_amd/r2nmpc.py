"""
r2nmpc.py -- host-side mirror of the reduced robustified NMPC (R2NMPC) on the batched solver.

Follows Model_Predictive_Controller/Reduced_Robustified_NMPC/Reduced_Robustified_NMPC_class.py:96-199 (set-up of
Sigma_0, W, B) and :249-406 (solve = nominal SQP-RTI step, then -- only if status == 0 -- covariance propagation
and constraint tightening for the NEXT solve). The tightening runs on the GPU (csrc/aux_kernels.hpp, K7).
"""
import numpy as np

from . import config as _config
from .solver import BatchedOcpSolver


def r2_setup(stds, Ts_MPC, coeff_sigma=0.5):
    """Sigma_0 (8x8), B W_disc B' (8x8) as the reference builds them (:106-141,148,177-184)."""
    w = np.asarray(stds, dtype=float)[2:6]                  # yaw, vlong, vlat, yawrate
    W_disc = Ts_MPC * np.diag(w) ** 2
    Sigma0 = (coeff_sigma * np.diag([1e-5, 1e-5, w[0], w[1], w[2], w[3], 1e-5, 1e-5])) ** 2
    B = np.zeros((8, 4)); B[2, 0] = B[3, 1] = B[4, 2] = B[5, 3] = 1.0
    return Sigma0, B @ W_disc @ B.T


class ReducedRobustifiedNMPC:
    def __init__(self, batch=1, N=38, dt=0.08, nsub=3, device=0, cfg=None, uph=None, stds=None):
        self.cfg = cfg or _config.default_config()
        m, veh = self.cfg["mpc"], self.cfg["veh"]
        self.uph = m["uncertainty_propagation_horizon"] if uph is None else uph
        self.Sigma0, self.BWB = r2_setup(m["stds"] if stds is None else stds, dt)
        self.delta_f_min, self.delta_f_max = veh["delta_f_min"], veh["delta_f_max"]
        self.acc_max = 1.0                                   # uh of the gg circle
        self.solver = BatchedOcpSolver(N=N, dt=dt, nsub=nsub, batch=batch, device=device, cfg=self.cfg, store_qp_in=True)
        self.solver.install_reference_ocp()
        self.N, self.batch = N, batch

    def solve(self):
        """One SQP-RTI step for every instance, then the tightening of lbx/ubx/uh for the next call
        (instances whose solve failed keep their previous bounds, like `if status == 0:` at :276)."""
        st = self.solver.solve()
        self.solver.r2_backoff(self.Sigma0, self.BWB, self.uph, self.delta_f_min, self.delta_f_max, self.acc_max)
        return st


class Reduced_Robustified_Nonlinear_Model_Predictive_Controller:
    """Host-side mirror of Reduced_Robustified_NMPC_class.py:37-470 (same constructor arguments, `solve(current_ref_traj)
    -> (u0, pred_X, stats)`, `set_initial_state`, `reset`, `update_cost_function_weights`): the nominal controller with
    `store_qp_in`, plus -- after every successful solve -- the covariance propagation and constraint tightening for the next
    one (:276-378; `ZoRo: False`, the shipped setting) as one device kernel (K7)."""

    def __init__(self, config_path=None, MPC_params_file=None, sim_main_params=None, X0_MPC=None, device=0, call_pattern="step"):
        from .nmpc import Nonlinear_Model_Predictive_Controller as _Nominal
        self._nom = _Nominal(config_path, MPC_params_file, sim_main_params, X0_MPC, device=device, store_qp_in=True,
                             call_pattern=call_pattern)
        n = self._nom
        m, veh = n.cfg["mpc"], n.cfg["veh"]
        self.cfg, self.MPC_params = n.cfg, n.MPC_params
        self.N, self.Tp, self.Ts, self.Ts_MPC, self.nx = n.N, n.Tp, n.Ts, n.Ts_MPC, n.nx
        self.model, self.constraint, self.ocp = n.model, n.constraint, n.ocp
        self.uncertainty_propagation_horizon = int(m["uncertainty_propagation_horizon"])
        self.Sigma0, self.BWB = r2_setup(m["stds"], self.Ts_MPC)
        self.delta_f_min, self.delta_f_max = veh["delta_f_min"], veh["delta_f_max"]
        self.acc_max = 1.0
        self.stats, self.pred_X = n.stats, n.pred_X
        self.WMPC = False
        self._attach()

    def _attach(self):
        """The tightening for the next solve (:276-378) is part of every solve of this capsule -- one more kernel behind the
        interior point method on the capsule's stream (tum_ocp_r2_attach), skipped for an instance whose solve failed
        (`if status == 0:` at :276) -- instead of a call and a round trip of its own after solve()."""
        self._nom._solver.r2_attach(self.Sigma0, self.BWB, self.uncertainty_propagation_horizon,
                                     self.delta_f_min, self.delta_f_max, self.acc_max)

    @property
    def acados_solver(self):
        """the nominal controller's solver (reading it flushes a pending initial state, nmpc._SolverHandle)"""
        return self._nom.acados_solver

    def solve(self, current_ref_traj):
        u0, pred_X, stats = self._nom.solve(current_ref_traj)          # (:379-381: time_tot includes the tightening -- it does: same stream)
        self.pred_X, self.stats = pred_X, stats
        return u0, pred_X, stats

    def set_initial_state(self, x0):
        self.x0 = x0
        self._nom.set_initial_state(x0)

    def reset(self, x0):
        self._nom.reset(x0)

    def reintialize_solver(self, X0_MPC, solver_generate_C_code=False, solver_build=False):
        self._nom.reintialize_solver(X0_MPC)
        self._attach()

    def update_cost_function_weights(self, params):
        self._nom.update_cost_function_weights(params)
