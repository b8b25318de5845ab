"""
Pins the CPU oracle (oracle/nmpc_oracle.c) against the reference's logged acados outputs
(tests/golden/*.npz, made by tests/golden/make_golden.py from
Learning_To_Adapt/SafeRL_WMPC/_baseline/F/*/*.npz). The reference has no tests of its
own; these logs are the only known answers for the hot path.
"""
import os

import numpy as np
import pytest

from oracle.oracle import OracleOcp, stm_f, rk4_sens, h_con


def _solve_kat(d, i):
    o = OracleOcp(38, 0.08, 3)
    o.set_weights(*d["params"][i])
    o.cold_start(d["x0"][i])
    y = d["yref"][i]
    o.set_yref(y[:, 0], y[:, 1], y[:, 2], y[:, 3])
    st = o.solve()
    return o, st


def test_kat0_all_52_logs(golden_dir):
    """Cold-start step 0 of every logged closed loop: u0, x1, cost."""
    d = np.load(os.path.join(golden_dir, "kat0.npz"))
    assert len(d["k"]) == 52
    for i in range(52):
        o, st = _solve_kat(d, i)
        assert st == 0
        # tolerance: 1e-6 relative (logs are reproduced to ~3e-8; north_star asks 1e-4)
        np.testing.assert_allclose(o.U[0], d["u0"][i], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(o.X[1], d["x1"][i], rtol=1e-6, atol=1e-8)
        assert abs(o.cost - d["cost"][i]) <= 1e-6 * abs(d["cost"][i])
        assert o.qp_iter <= 50


@pytest.mark.parametrize("name,tol", [("replay_lvms_0_0_450.npz", 2e-6), ("replay_monteblanco_0_0_400.npz", 2e-5)])
def test_sequential_replay(golden_dir, name, tol):
    """Warm-started RTI sequence (iterate carried over un-shifted): exercises the
    linearised lower bound of h, active soft slacks (lvms step 55) and the hard-active
    h<=1 rows (lvms 300-326). Tolerance absolute on (u0, x1); north_star asks 1e-4 rel."""
    d = np.load(os.path.join(golden_dir, name))
    o = OracleOcp(38, 0.08, 3)
    o.set_weights(*d["params"])
    worst = 0.0
    for i in range(len(d["x0"])):
        if i == 0:
            o.cold_start(d["x0"][0])
        else:
            o.x0[:] = d["x0"][i]
        y = d["yref"][i]
        o.set_yref(y[:, 0], y[:, 1], y[:, 2], y[:, 3])
        assert o.solve() == 0
        worst = max(worst, np.abs(o.U[0] - d["u0"][i]).max(), np.abs(o.X[1] - d["x1"][i]).max())
        assert abs(o.cost - d["cost"][i]) <= 1e-4 * max(1.0, abs(d["cost"][i]))
    assert worst < tol


def hard_window_check(golden_dir, solve_step):
    """Shared by the oracle test (here) and the HIP test (test_gpu_parity.py): replays every window of replay_hard.npz --
    the logged closed loops of the weight sets 2, 13, 16, 18 around the first QP solve that acados stopped at its
    50-iteration cap -- through `solve_step(key, i, params, x0, yref) -> (u0, x1, qp_iter, converged)` and holds every solve
    to what acados logged, split by what acados itself reports about that solve:
      * a solve is COMPARABLE when acados converged on it (qp_iter < 50) and on the 25 solves before it (a capped solve
        leaves a different warm start behind; the RTI sequence forgets it within about 20 steps) and, in the window that
        starts in the middle of a log, after 20 warm-up steps: those must match to 2e-4 (observed <= 7e-5; north_star 1e-4
        relative) -- this is every solve of the converged stretch before the first capped one, hundreds per window;
      * on the CAPPED solves themselves this solver must have converged (status 0, KKT residuals at tolerance, <= 20
        iterations): the deviation there is acados' unfinished QP step, not this solver's.
    Returns per window (n comparable, worst comparable error, worst error at / after capped solves)."""
    from tum_control_amd.planner import load_track, planner_emulator, yref_from_ref
    g = np.load(os.path.join(golden_dir, "replay_hard.npz"))
    out = {}
    for track, k, start in zip(g["track"], g["k"], g["start"]):
        key = f"{track}_{k}"
        tr = load_track(str(track))
        x0s, poses, aq = g[key + "_x0"], g[key + "_pose"], g[key + "_qp_iter"]
        n = len(x0s)
        capped = aq >= 50
        assert capped.any() or start > 0
        tainted = np.zeros(n, bool)
        for c in np.nonzero(capped)[0]:
            tainted[c:c + 26] = True
        warm = np.zeros(n, bool)
        if start > 0:
            warm[:20] = True
        worst_ok = worst_cap = 0.0
        for i in range(n):
            _, ref = planner_emulator(tr, poses[i], 39, 3.04, True)
            u0, x1, it, converged = solve_step(key, i, g["params"][int(k)], x0s[i], yref_from_ref(ref, 38))
            assert converged and it <= 20, (key, i, it)
            dx = x1 - g[key + "_x1"][i]
            dx[2] = (dx[2] + np.pi) % (2 * np.pi) - np.pi            # the logs hold yaw mod 2 pi
            e = max(np.abs(u0 - g[key + "_u0"][i]).max(), np.abs(dx).max())
            if warm[i]:
                continue
            if tainted[i]:
                worst_cap = max(worst_cap, e)
            else:
                worst_ok = max(worst_ok, e)
                assert e < 2e-4, (key, i, e, int(aq[i]))
        out[key] = (int((~tainted & ~warm).sum()), worst_ok, worst_cap)
    return out


def test_hard_weight_sets_per_solve(golden_dir):
    """Per-solve parity on the weight sets where acados hit its QP iteration cap."""
    state = {}

    def step(key, i, params, x0, yref):
        if i == 0:
            o = state[key] = OracleOcp(38, 0.08, 3)
            o.set_weights(*params)
            o.cold_start(x0)
        o = state[key]
        o.x0[:] = x0
        o.yref[:] = yref
        st = o.solve()
        return o.U[0].copy(), o.X[1].copy(), o.qp_iter, st == 0 and o.res.max() < 1e-6

    res = hard_window_check(golden_dir, step)
    assert sum(r[0] for r in res.values()) > 1500                       # comparable solves held to 2e-4
    # where the logged loops part from this solver it is at a capped solve: on several windows the deviation at / right
    # after the first capped solve is orders of magnitude above anything on the converged stretches (some capped solves
    # were almost converged and deviate little: not every window shows it)
    assert sum(r[2] > 1e-3 and r[2] > 40 * r[1] for r in res.values()) >= 2, res


def test_jacobians_finite_difference():
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = np.array([rng.normal(0, 50), rng.normal(0, 50), rng.uniform(-7, 7), rng.uniform(3, 38),
                      rng.normal(0, 0.3), rng.normal(0, 0.1), rng.normal(0, 0.05), rng.uniform(-3, 2.5)])
        u = rng.normal(0, 1, 2)
        f0, Jx, Ju = stm_f(x, u)
        for j in range(8):
            e = np.zeros(8); e[j] = 1e-6 * max(1.0, abs(x[j]))
            fd = (stm_f(x + e, u)[0] - stm_f(x - e, u)[0]) / (2 * e[j])
            np.testing.assert_allclose(Jx[:, j], fd, rtol=2e-6, atol=2e-7)
        xn, A, B = rk4_sens(x, u, 0.08, 3)
        for j in range(8):
            e = np.zeros(8); e[j] = 1e-6 * max(1.0, abs(x[j]))
            fd = (rk4_sens(x + e, u, 0.08, 3)[0] - rk4_sens(x - e, u, 0.08, 3)[0]) / (2 * e[j])
            np.testing.assert_allclose(A[:, j], fd, rtol=2e-6, atol=2e-7)
        for j in range(2):
            e = np.zeros(2); e[j] = 1e-6
            fd = (rk4_sens(x, u + e, 0.08, 3)[0] - rk4_sens(x, u - e, 0.08, 3)[0]) / 2e-6
            np.testing.assert_allclose(B[:, j], fd, rtol=2e-6, atol=2e-7)
        h0, gh = h_con(x)
        for j in range(8):
            e = np.zeros(8); e[j] = 1e-6 * max(1.0, abs(x[j]))
            fd = (h_con(x + e)[0] - h_con(x - e)[0]) / (2 * e[j])
            assert abs(gh[j] - fd) <= 2e-6 * max(1.0, abs(fd))


def test_model_structure():
    """delta and a are pure integrators of the inputs (rows 6,7 of A,B)."""
    x = np.array([1.0, 2.0, 0.5, 20.0, 0.1, 0.05, 0.02, 0.7]); u = np.array([0.3, -0.1])
    xn, A, B = rk4_sens(x, u, 0.08, 3)
    np.testing.assert_allclose(xn[6], x[6] + 0.08 * u[1], rtol=0, atol=1e-15)
    np.testing.assert_allclose(xn[7], x[7] + 0.08 * u[0], rtol=0, atol=1e-15)
    np.testing.assert_allclose(A[6], np.eye(8)[6], atol=1e-15)
    np.testing.assert_allclose(A[7], np.eye(8)[7], atol=1e-15)
    np.testing.assert_allclose(B[6], [0, 0.08], atol=1e-15)
    np.testing.assert_allclose(B[7], [0.08, 0], atol=1e-15)


@pytest.mark.parametrize("track,k", [("monteblanco", 0), ("lvms", 7)])
def test_oracle_closed_loop_full_length(golden_dir, track, k):
    """The oracle in a CLOSED loop (planner -> oracle solve -> plant -> moving-average estimator, all on the host) for the
    complete 5499 logged control steps of one weight set per track, against every 25th logged plant state: pins the oracle
    together with the planner / plant / estimator restatements over a whole lap and a half (SURVEY 8(c) KAT-replay)."""
    from tum_control_amd import config
    from tum_control_amd.closed_loop import plant_step, MovingAverageEstimator
    from tum_control_amd.planner import load_track, planner_emulator
    g = np.load(os.path.join(golden_dir, f"closed_loop_{track}_full_sub25.npz"))
    sub, n = int(g["sub"]), 5499
    cfg = config.default_config()
    tr = load_track(track)
    o = OracleOcp(38, 0.08, 3)
    o.set_weights(*g["params"][k])
    x_mpc = np.array([tr[0, 0], tr[0, 1], np.mod(tr[0, 2], 2 * np.pi), tr[0, 3], 0, 0, 0, 0.0])
    x_sim = x_mpc[None, :7].copy()
    pose = x_mpc[:2].copy()
    est = MovingAverageEstimator(1)
    o.cold_start(x_mpc)
    C = [x_sim[0].copy()]
    for i in range(n):
        _, ref = planner_emulator(tr, pose, 39, 3.04, True)
        o.set_yref(ref[:, 0], ref[:, 1], ref[:, 2], ref[:, 3])
        assert o.solve() == 0
        x_sim = plant_step(x_sim, np.array([o.X[1][7]]), np.array([o.U[0][1]]), cfg)
        pose = x_sim[0, :2].copy()
        o.x0[:] = est(np.concatenate([x_sim, [[o.X[1][7]]]], axis=1))[0]
        if (i + 1) % sub == 0:
            C.append(x_sim[0].copy())
    C = np.array(C)
    ref_c = g["CiLX"][k][:len(C)]
    ep = np.hypot(C[:, 0] - ref_c[:, 0], C[:, 1] - ref_c[:, 1])
    # observed: median 1.7e-8 m, max 2.7e-6 m (monteblanco/0), 3.3e-7 m (lvms/7) over 110 s of driving
    assert np.median(ep) < 1e-6 and ep.max() < 1e-4, (np.median(ep), ep.max())
    assert np.abs(C[:, 3] - ref_c[:, 3]).max() < 1e-4


def test_full_logs_per_solve_gate(golden_dir):
    """ALL 52 complete logged acados closed loops (2 x 26 x 5499 solves), every solve against the log at north_star's 1e-4
    relative (tests/golden/replay_full_logs.py: definitions, comparable solves, the evidence kept for every exception).
    Where the reference's logs are present (the build container) the replay is re-run -- about a minute on eight cores -- and
    must reproduce the committed report; elsewhere the gate is asserted on the committed report."""
    import json
    import sys
    sys.path.insert(0, golden_dir)
    import replay_full_logs as R
    committed = json.load(open(os.path.join(golden_dir, "full_replay_report.json")))
    summary = R.gate(committed["logs"])
    assert committed["tol"] == 1e-4 and len(committed["logs"]) == 52
    ncomp = sum(r["n_comparable"] for r in committed["logs"])
    nexc = sum(len(r["exceptions"]) for r in committed["logs"])
    assert ncomp > 280000 and nexc == 13 <= R.MAX_EXC, summary          # (ratcheted to the committed report, round 6)
    assert max(r["worst_comparable"] for r in committed["logs"]) <= R.WORST_BOUND == 6.5e-3
    # the typical agreement is seven orders of magnitude below the gate
    assert np.median([r["median_comparable"] for r in committed["logs"]]) < 2e-8
    if not R.available():
        return
    live = R.run()
    R.gate(live)
    for a, b in zip(live, committed["logs"]):
        assert (a["track"], a["k"], a["n_comparable"]) == (b["track"], b["k"], b["n_comparable"])
        assert [e["step"] for e in a["exceptions"]] == [e["step"] for e in b["exceptions"]], (a["track"], a["k"])
        assert abs(a["worst_comparable"] - b["worst_comparable"]) <= 1e-6 + 0.05 * b["worst_comparable"]


def test_forced_iteration_count():
    """ipm_opts.iter_force (what the GPU tests use on instances whose termination test sits on the tolerance edge): with the
    oracle's own count imposed the result is bit-identical, one iteration more moves it at the 1e-6 level and no further."""
    from tum_control_amd import config
    from tum_control_amd.workloads import nominal_batch
    m = config.MPC
    o = OracleOcp(40, 0.08, 3)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    x0, yref = nominal_batch(24, N=40)
    u, X1, st = o.solve_batch_cold(x0, yref, 4)
    it = st[:, 1].astype(int)
    u2, X2, st2 = o.solve_batch_cold(x0, yref, 4, force_iter=it)
    assert np.array_equal(u, u2) and np.array_equal(X1, X2) and np.array_equal(st, st2)
    u3, _, st3 = o.solve_batch_cold(x0, yref, 4, force_iter=it + 1)
    assert np.array_equal(st3[:, 1], it + 1) and (st3[:, 2] == 0).all()
    assert 0 < np.abs(u3 - u).max() < 5e-5
