"""
planner.py -- local reference extraction (the producer of yref, one step before the hot path).

Own restatement of the behaviour of Utils/MPC_sim_utils.py:137-194 (PlannerEmulator): nearest
waypoint, walk forward until the travel time exceeds Tp, resample the extracted points to N+1
equally spaced (in index) samples, with the 2*pi wrap special-case for the reference yaw.
Checked against golden vectors captured from the reference (tests/golden/planner.npz).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "reftraj.npz")
TRACKS = ("monteblanco", "lvms", "modena")


def load_track(name):
    """(n,4) array [pos_x, pos_y, ref_yaw, ref_v] of one of the shipped race lines."""
    with np.load(_DATA) as d:
        if name not in d.files:
            raise KeyError(f"unknown track '{name}' (have {d.files})")
        return d[name].copy()


def planner_emulator(track, pose_xy, n_points, Tp, loop_circuit=True):
    """Returns (closest_index, ref) with ref an (n_points, 4) array [pos_x, pos_y, ref_yaw, ref_v]."""
    px, py, yaw, v = track[:, 0], track[:, 1], track[:, 2], track[:, 3]
    n = len(px)
    d2 = (px - pose_xy[0]) ** 2 + (py - pose_xy[1]) ** 2
    i0 = int(np.argmin(d2))
    idx = [i0]
    T = 0.0
    while T <= Tp:
        cur = idx[-1]
        nxt = cur + 1
        if nxt >= n:
            if not loop_circuit:
                break
            nxt = 0
        idx.append(nxt)
        T += float(np.hypot(px[nxt] - px[cur], py[nxt] - py[cur])) / v[nxt]
    idx = np.asarray(idx)
    m = len(idx)
    seg = track[idx]
    if m == n_points:
        return i0, seg.copy()
    xs = np.linspace(0.0, m - 1, n_points)
    xp = np.arange(m)
    out = np.empty((n_points, 4))
    for c in range(4):
        out[:, c] = np.interp(xs, xp, seg[:, c])
    # the reference yaw lives in [0, 2pi): interpolate across a wrap on the unwrapped signal
    if (np.abs(np.diff(seg[:, 2])) > np.deg2rad(250)).any():
        out[:, 2] = np.mod(np.interp(xs, xp, np.unwrap(seg[:, 2], period=2 * np.pi)), 2 * np.pi)
    return i0, out


def yref_from_ref(ref, N):
    """(N+1, 6) yref block for the solver: [x, y, yaw, v, 0, 0] (NMPC_class.py:169-180)."""
    y = np.zeros((N + 1, 6))
    y[:, :4] = ref[:N + 1]
    return y
