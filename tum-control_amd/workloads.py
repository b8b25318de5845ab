"""
workloads.py -- synthetic batches for the BASELINE.json configs.

Every instance is one nominal NMPC OCP: an initial state x0 (a pose on a race line plus a
state-estimation style perturbation, Config/EDGAR/sim_main_params.yaml:54-60) and the planner's
local reference yref for that pose. All generated on the host with fixed seeds.
"""
import numpy as np

from . import config as _config
from .planner import load_track, planner_emulator, yref_from_ref


def _pose_state(track, i):
    return np.array([track[i, 0], track[i, 1], np.mod(track[i, 2], 2 * np.pi), track[i, 3], 0.0, 0.0, 0.0, 0.0])


def nominal_batch(batch, N=40, dt=0.08, track_name="monteblanco", stride=37, seed=1234, offset=0, noise=None):
    """Config 2: perturbed-x0 fan-out. Pose index (stride*b) mod n_track, x0 = pose + N(0, diag(w)^2),
    vlong clipped >= 1; yref from the planner at the perturbed position. Returns x0 (B,8), yref (B,N+1,6)."""
    tr = load_track(track_name)
    n = len(tr)
    w = np.asarray(_config.SIM["w_state_estimation"] if noise is None else noise, dtype=float)
    rng = np.random.default_rng(seed)
    x0 = np.zeros((batch, 8)); yref = np.zeros((batch, N + 1, 6))
    Tp = N * dt
    for b in range(batch):
        i = (stride * (b + offset)) % n
        x = _pose_state(tr, i) + w * rng.standard_normal(8)
        x[3] = max(x[3], 1.0)
        _, ref = planner_emulator(tr, x[:2], N + 1, Tp, True)
        x0[b] = x
        yref[b] = yref_from_ref(ref, N)
    return x0, yref


def scenario_batch(n_poses, x0_offsets, N=40, dt=0.08, track_name="monteblanco", pose_stride=None):
    """Configs 3/4: scenario fan-out. Each pose contributes len(x0_offsets)+1 instances (nominal +
    scenarios x0 + offset_s) that share one yref. Returns x0 (P*(S+1),8), yref (P*(S+1),N+1,6), group size."""
    tr = load_track(track_name)
    n = len(tr)
    S1 = len(x0_offsets) + 1
    x0 = np.zeros((n_poses * S1, 8)); yref = np.zeros((n_poses * S1, N + 1, 6))
    for p in range(n_poses):
        i = (p * n) // n_poses if pose_stride is None else (pose_stride * p) % n
        x = _pose_state(tr, i)
        _, ref = planner_emulator(tr, x[:2], N + 1, N * dt, True)
        y = yref_from_ref(ref, N)
        x0[p * S1] = x
        x0[p * S1 + 1:(p + 1) * S1] = x[None, :] + x0_offsets
        yref[p * S1:(p + 1) * S1] = y[None]
    return x0, yref, S1
