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


# ---------------------------------------------------------------------------------------------------------------------
# The BASELINE.json configs as (scenario-)GROUP ranges. A group is the unit that must not be split across GPUs: one
# instance for configs 2 and 5, one pose with all its scenarios (nominal + draws, sharing yref) for configs 3 and 4
# ("scenario groups are not split across ranks"). Every generator takes a range [g_lo, g_hi) of GLOBAL
# group indices of a job with `groups_total` groups and draws its random numbers from per-group streams keyed
# (seed, global group index), so the union of the shards of any world size IS the unsharded batch
# (tests/test_host_logic.py::test_group_sharding_*).
CONFIGS = {
    2: dict(name="nominal NMPC, perturbed x0", track="monteblanco", group=1, groups_per_gpu=4096, seed=1234, solves_per_step=1),
    3: dict(name="SNMPC sigma-point scenarios (nominal + 15 Hammersley points per pose) + PCE moments", track="monteblanco",
            group=16, groups_per_gpu=1024, seed=0, solves_per_step=1),
    4: dict(name="SNMPC Monte-Carlo scenarios (nominal + 15 i.i.d. draws per pose)", track="lvms", group=16, groups_per_gpu=1024,
            seed=4321, solves_per_step=1),
    5: dict(name="R2NMPC: solve, covariance back-off (K7), solve with the tightened bounds", track="modena", group=1,
            groups_per_gpu=4096, seed=777, solves_per_step=2),
}


def _pose_yref(tr, x, N, dt):
    _, ref = planner_emulator(tr, x[:2], N + 1, N * dt, True)
    return yref_from_ref(ref, N)


def perturbed_pose_groups(g_lo, g_hi, N=40, dt=0.08, track_name="modena", stride=37, seed=777, noise=None, shift=0):
    """Configs 2-style instances with shard-invariant random streams (used for config 5, SURVEY 8(d) row 5:
    "Modena poses + same noise"): instance b = pose (stride*b) mod n + N(0, diag(w)^2) from the stream (seed, b)."""
    tr = load_track(track_name)
    n = len(tr)
    w = np.asarray(_config.SIM["w_state_estimation"] if noise is None else noise, dtype=float)
    B = g_hi - g_lo
    x0 = np.zeros((B, 8)); yref = np.zeros((B, N + 1, 6))
    for j, b in enumerate(range(g_lo, g_hi)):
        x = _pose_state(tr, (stride * b + shift) % n) + w * np.random.default_rng([seed, b]).standard_normal(8)
        x[3] = max(x[3], 1.0)
        x0[j] = x
        yref[j] = _pose_yref(tr, x, N, dt)
    return x0, yref


def sigma_point_groups(g_lo, g_hi, groups_total, offsets, N=40, dt=0.08, track_name="monteblanco", shift=0):
    """Config 3: pose p = race-line point (p * n) // groups_total; its group is the nominal instance
    followed by pose + offsets[s] (the Hammersley sigma points scaled by the stds, snmpc.x0_offsets); one yref per group.
    Returns the POSE states (G, 8), the group yref (G, N+1, 6) and the expanded x0 (G*S1, 8), yref (G*S1, N+1, 6)."""
    tr = load_track(track_name)
    n = len(tr)
    offsets = np.asarray(offsets, dtype=float).reshape(-1, 8)
    S1 = len(offsets) + 1
    G = g_hi - g_lo
    pose = np.zeros((G, 8)); yg = np.zeros((G, N + 1, 6))
    for j, p in enumerate(range(g_lo, g_hi)):
        pose[j] = _pose_state(tr, ((p * n) // groups_total + shift) % n)
        yg[j] = _pose_yref(tr, pose[j], N, dt)
    x0 = np.repeat(pose, S1, axis=0)
    x0.reshape(G, S1, 8)[:, 1:] += offsets[None]
    return pose, yg, x0, np.repeat(yg, S1, axis=0)


def monte_carlo_groups(g_lo, g_hi, N=40, dt=0.08, track_name="lvms", pose_stride=7, draws=15, stds=None, seed=4321, shift=0):
    """Config 4: pose p = race-line point (7 p) mod n (LVMS), its group is the nominal instance followed
    by `draws` scenarios pose + N(0, diag(stds)^2), i.i.d. per scenario from the stream (seed, p); one yref per group.
    (The scenario generator is compute_x0dist, stochastic_mpc_utils.py:78-91, with random instead of Hammersley samples:
    the reference has no Monte-Carlo code)"""
    tr = load_track(track_name)
    n = len(tr)
    stds = np.asarray(_config.MPC["stds"] if stds is None else stds, dtype=float)
    S1 = draws + 1
    G = g_hi - g_lo
    x0 = np.zeros((G * S1, 8)); yref = np.zeros((G * S1, N + 1, 6))
    for j, p in enumerate(range(g_lo, g_hi)):
        x = _pose_state(tr, (pose_stride * p + shift) % n)
        y = _pose_yref(tr, x, N, dt)
        x0[j * S1] = x
        x0[j * S1 + 1:(j + 1) * S1] = x[None] + stds[None] * np.random.default_rng([seed, p]).standard_normal((draws, 8))
        yref[j * S1:(j + 1) * S1] = y[None]
    return x0, yref


def config_groups(config_id, g_lo, g_hi, groups_total, N=40, dt=0.08, variant=0):
    """x0 (n, 8), yref (n, N+1, 6) of the groups [g_lo, g_hi) of BASELINE configs[config_id - 1], group size.
    variant = 0 is the configuration as BASELINE / SURVEY 8(d) define it; variant k > 0 is a FRESH batch of the same
    workload -- the poses moved along the race line by 3 k points and other random streams (seed + 1000 k) -- which
    bench.py rotates through so that no step sees the batch of the step before."""
    c = CONFIGS[config_id]
    shift, dseed = 3 * int(variant), 1000 * int(variant)
    if config_id == 2:
        # the round-1 benchmark batch (sequential stream): identical to nominal_batch(batch, offset=g_lo) for g_lo == 0
        x0, yref = nominal_batch(g_hi - g_lo, N=N, dt=dt, track_name=c["track"], stride=37,
                                 seed=c["seed"] + dseed + (g_lo // max(g_hi - g_lo, 1)), offset=g_lo + 5 * int(variant))
    elif config_id == 3:
        from .snmpc import hammersley_normal, x0_offsets
        off = x0_offsets(hammersley_normal(15, 3), _config.MPC["stds"])
        _, _, x0, yref = sigma_point_groups(g_lo, g_hi, groups_total, off, N=N, dt=dt, track_name=c["track"], shift=shift)
    elif config_id == 4:
        x0, yref = monte_carlo_groups(g_lo, g_hi, N=N, dt=dt, track_name=c["track"], seed=c["seed"] + dseed, shift=shift)
    elif config_id == 5:
        x0, yref = perturbed_pose_groups(g_lo, g_hi, N=N, dt=dt, track_name=c["track"], seed=c["seed"] + dseed, shift=shift)
    else:
        raise ValueError(f"no batch workload for config {config_id}")
    return x0, yref, c["group"]
