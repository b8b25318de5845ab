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
        ok = self.solver.get_stats("status") == 0
        if ok.all():
            self.solver.r2_backoff(self.Sigma0, self.BWB, self.uph, self.delta_f_min, self.delta_f_max, self.acc_max)
        return st
