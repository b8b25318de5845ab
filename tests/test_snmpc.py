"""
The coupled SNMPC OCP (Stochastic_NMPC/SNMPC_acados_settings.py, pred_model_dynamic_disc.py).

The reference holds no solver outputs of this OCP with a recorded configuration (ACC24 logs are "weak"), so
the SOLVER OUTPUTS of the oracle's SNMPC section are parity-unpinned against acados. What pins it here:
  * its model functions (stacked dynamics, cost output, chance constraint) and problem data against the reference's own
    exported OCP, acados_ocp_SNMPC.json (snmpc_expr.npz / snmpc_json.npz),
  * with stop_flag = 1 everywhere it must reduce to the (golden-pinned) nominal restatement with one RK4 step,
  * its constraint rows and defects must agree with finite differences of an independent numpy rollout of
    pred_model_dynamic_disc.py (written from the reference text, sharing only the single-track RK4 step),
and the HIP path is then held to the oracle.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc


def _pce():
    from tum_control_amd import snmpc as snm
    stds = np.array([0, 0, 0, .8, .35, .035, 0, 0])
    w = snm.hammersley_normal(10, 3)
    A = snm.pce_matrix(w, snm.alpha_generation(3, 2))
    return snm, stds, w, A


def _kat(golden_dir, i=0):
    d = np.load(os.path.join(golden_dir, "kat0.npz"))
    return d["x0"][i], d["yref"][i], d["params"][i]


def test_oracle_reduces_to_nominal(golden_dir):
    """uph = 0 (stop_flag = 1 on every stage): the nominal copy runs its own RK4 step and the constraint is h(nominal);
    at vt = 0 (cold start of the logged poses) |v| = vl, so the first solve equals the nominal OCP with nsub = 1."""
    snm, stds, w, A = _pce()
    x0, yref, p = _kat(golden_dir)
    N = yref.shape[0] - 1
    o = orc.OracleSnmpcOcp(N=N, dt=0.08, Apce=A, uph=0)
    o.set_weights(*p); o.yref[:, :4] = yref; o.cold_start(snm.compute_x0dist(x0, w, stds))
    n = orc.OracleOcp(N=N, dt=0.08, nsub=1)
    n.set_weights(*p); n.yref[:, :4] = yref; n.cold_start(x0)
    assert o.solve() == 0 and n.solve() == 0
    np.testing.assert_allclose(o.U, n.U, rtol=0, atol=1e-13)
    np.testing.assert_allclose(o.X[:, 0], n.X, rtol=0, atol=1e-11)
    assert abs(o.cost - n.cost) < 1e-6 * n.cost          # (|v| vs vl at the new iterate)


def _rollout(x0s, U, A, uph, N, dt, kappa):
    """pred_model_dynamic_disc.py:170-212 + SNMPC_acados_settings.py:100-133,187 in numpy: stacked states and h per stage."""
    ns = x0s.shape[0] - 1
    X = np.zeros((N + 1, ns + 1, 8)); X[0] = x0s
    hv = np.zeros(N + 1)
    for k in range(N):
        stop = k >= uph
        for i in range(1, ns + 1):
            X[k + 1, i] = X[k, i] if stop else orc.rk4_sens(X[k, i], U[k], dt, 1)[0]
        X[k + 1, 0] = orc.rk4_sens(X[k, 0], U[k], dt, 1)[0] if stop else A[0] @ X[k + 1, 1:]
    for k in range(1, N + 1):
        if k >= uph:
            hv[k] = orc.h_con_vabs(X[k, 0])[0]
        else:
            c = A @ np.array([orc.h_con_vabs(X[k, i])[0] for i in range(1, ns + 1)])
            hv[k] = c[0] + kappa * np.sqrt((c[1:] ** 2).sum())
    return X, hv


def test_oracle_rows_against_finite_differences(golden_dir):
    snm, stds, w, A = _pce()
    x0, yref, p = _kat(golden_dir)
    x0 = x0.copy(); x0[7] = 0.8; x0[5] = 0.12; x0[4] = 0.3        # a, r, vt away from zero: every gradient entry is alive
    N, uph, dt = 12, 5, 0.08
    rng = np.random.default_rng(3)
    U = np.stack([rng.normal(0, 1.0, N), rng.normal(0, 0.05, N)], axis=1)
    xs = snm.compute_x0dist(x0, w, stds)
    o = orc.OracleSnmpcOcp(N=N, dt=dt, Apce=A, uph=uph)
    o.set_weights(*p); o.yref[:, :4] = yref[:N + 1]
    X, hv = _rollout(xs, U, A, uph, N, dt, 0.5)
    o.x0[:] = xs; o.X[:] = X; o.U[:] = U                          # a dynamically feasible iterate: all defects vanish
    o.set_iter_max(1)
    _, qp = o.solve_debug()
    C, d = qp["C"], qp["d"]
    np.testing.assert_allclose(d[N + 1::2], hv[1:], rtol=0, atol=1e-12)          # h rows: value at the iterate
    np.testing.assert_allclose(d[N::2], X[1:, 0, 6], rtol=0, atol=1e-12)         # steering-angle rows
    eps = 1e-6
    J = np.zeros((N, 2 * N))
    Jd = np.zeros((N, 2 * N))
    for j in range(2 * N):
        Up = U.copy(); Up[j // 2, j % 2] += eps
        Um = U.copy(); Um[j // 2, j % 2] -= eps
        Xp, hp = _rollout(xs, Up, A, uph, N, dt, 0.5)
        Xm, hm = _rollout(xs, Um, A, uph, N, dt, 0.5)
        J[:, j] = (hp[1:] - hm[1:]) / (2 * eps)
        Jd[:, j] = (Xp[1:, 0, 6] - Xm[1:, 0, 6]) / (2 * eps)
    np.testing.assert_allclose(C[N + 1::2], J, rtol=2e-6, atol=2e-8)
    np.testing.assert_allclose(C[N::2], Jd, rtol=1e-6, atol=1e-9)
    assert np.abs(J[:uph - 1]).max() > 1e-3                       # the chance rows are not trivially zero


def test_oracle_h_vabs_gradient():
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = np.array([0, 0, 0, rng.uniform(5, 40), rng.uniform(-2, 2), rng.uniform(-.3, .3), 0, rng.uniform(-3, 3)])
        x[3] = min(x[3], 11.5) if _ % 4 == 0 else x[3]            # some points on the sloped part of the ax table
        h, g = orc.h_con_vabs(x)
        for i in (3, 4, 5, 7):
            e = np.zeros(8); e[i] = 1e-6
            fd = (orc.h_con_vabs(x + e)[0] - orc.h_con_vabs(x - e)[0]) / 2e-6
            assert abs(fd - g[i]) < 1e-6 * max(1.0, abs(fd))


def test_wrapper_rejects_wrong_parameter_dimension():
    """host logic only: the dimension check of set(stage, 'p', ...) runs before any device call (acados' message); the
    meaning of the parameter vector is handled by the C-ABI (test_gpu_per_stage_parameters)"""
    from tum_control_amd.solver import CoupledSnmpcSolver
    s = CoupledSnmpcSolver.__new__(CoupledSnmpcSolver)
    s.Apce = np.arange(6.0).reshape(2, 3); s.L, s.ns, s.uph = 2, 3, 4
    good = np.concatenate((s.Apce.flatten(), [0.8], [0.0]))
    with pytest.raises(Exception, match="mismatching dimension"):
        s.set(2, "p", good[:-1])


def test_reintialize_solver_keeps_the_controller_configuration(monkeypatch):
    """SNMPC_class.py:274-281 rebuilds the solver with the SAME Tp / N / Q / R / penalties / PCE set-up (main.py:59-61 calls
    it after every failed solve). CPU test with a recording stand-in for the device solver."""
    from tum_control_amd import snmpc
    built = []

    class FakeSolver:
        def __init__(self, **kw):
            self.kw = kw; self.calls = []; built.append(self)

        def install_reference_ocp(self, **kw):
            self.ocp = kw

        def constraints_set(self, *a):
            self.calls.append(("constraints_set", a[0], a[1], np.array(a[2])))

        def set(self, stage, field, v):
            self.calls.append(("set", stage, field, np.array(v)))

        def cold_start(self):
            self.calls.append(("cold_start",))

        def set_x0_offsets(self, o):
            self.calls.append(("set_x0_offsets", np.array(o)))

    monkeypatch.setattr(snmpc, "CoupledSnmpcSolver", FakeSolver)
    cfg = snmpc._config.default_config()
    cfg["mpc"].update(q_lon=7.5, r_jerk=11.0, L1_pen=55.0, n_samples=12, uncertainty_propagation_horizon=9, gamma=0.9,
                      stds=[0, 0, 0, 0.5, 0.2, 0.02, 0, 0])
    monkeypatch.setattr(snmpc._config, "default_config", lambda: cfg)
    x0 = np.array([1.0, 2.0, 0.3, 20.0, 0.0, 0.0, 0.0, 0.0])
    c = snmpc.Stochastic_Nonlinear_Model_Predictive_Controller(sim_main_params=dict(Tp=2.0, Ts=0.02, Ts_MPC=0.1), X0_MPC=x0)
    assert c.N == 20 and len(built) == 1
    monkeypatch.setattr(snmpc._config, "default_config", lambda: (_ for _ in ()).throw(AssertionError("defaults must not be re-read")))
    x1 = x0 + 0.5
    c.reintialize_solver(x1)
    assert len(built) == 2 and c.acados_solver is built[1]
    a, b = built
    assert b.kw["N"] == 20 and b.kw["uph"] == 9 and b.kw["gamma"] == 0.9 and b.kw["cfg"] is cfg
    assert b.kw["Apce"].shape == (10, 12) and np.array_equal(b.kw["Apce"], a.kw["Apce"])
    assert abs(b.kw["dt"] - 0.1) < 1e-15
    for k in ("Q", "R", "Qe"):
        assert np.array_equal(b.ocp[k], a.ocp[k])
    assert b.ocp["Q"][0, 0] == 7.5 and b.ocp["R"][0, 0] == 11.0 and b.ocp["L1"] == 55.0
    # cold start at the sample states of the NEW x0, every stage parameterised like at construction
    lbx = [cl for cl in b.calls if cl[0] == "constraints_set" and cl[2] == "lbx"]
    np.testing.assert_allclose(lbx[0][3].reshape(13, 8)[0], x1)
    np.testing.assert_allclose(lbx[0][3].reshape(13, 8)[1:, 3] - x1[3], 0.5 * c.w_samples[0], atol=1e-14)
    ps = [cl for cl in b.calls if cl[0] == "set" and cl[2] == "p"]
    assert len(ps) == 21 and [int(q[3][-1]) for q in ps] == [0] * 9 + [1] * 12
    assert ("cold_start",) in b.calls
    # the sample offsets for the 8-value state of a control step (set_initial_state rides with the step): stds (.) w_s
    offs = [cl for cl in b.calls if cl[0] == "set_x0_offsets"]
    assert len(offs) == 1 and offs[0][1].shape == (12, 8)
    np.testing.assert_allclose(offs[0][1][:, 3], 0.5 * c.w_samples[0], atol=1e-14)
    assert (offs[0][1][:, [0, 1, 2, 6, 7]] == 0).all()
    # set_initial_state no longer talks to the solver: the state is pending until the next solve() (or a flush)
    n_calls = len(b.calls)
    c.set_initial_state(x1 + 0.25)
    assert len(b.calls) == n_calls and np.array_equal(c._x0_pending, x1 + 0.25)
    c._flush_x0()
    lbx2 = [cl for cl in b.calls[n_calls:] if cl[0] == "constraints_set" and cl[2] == "lbx"]
    np.testing.assert_allclose(lbx2[0][3].reshape(13, 8)[0], x1 + 0.25)
    assert c._x0_pending is None


def test_problem_data_matches_exported_ocp(golden_dir):
    """acados_ocp_SNMPC.json (the reference's exported OCP) against the constants this build installs: dimensions of the
    stacked problem, W = 0.01 blockdiag(Q, R), slack penalties per class, bounds, solver options (CPU: configuration only)."""
    import json
    from tum_control_amd import config
    from tum_control_amd.solver import TumOcpDesc, make_desc
    g = np.load(os.path.join(golden_dir, "snmpc_json.npz"))
    dims, opts = json.loads(str(g["dims"])), json.loads(str(g["opts"]))
    cfg = config.default_config(); m, veh = cfg["mpc"], cfg["veh"]
    ns = m["n_samples"]
    L = len(_pce()[0].alpha_generation(int(np.count_nonzero(m["stds"])), m["expansion_degree"]))
    assert dims["nx"] == 8 * (ns + 1) and dims["nu"] == 2 and dims["np"] == L * ns + 2 == int(g["n_param"])
    assert dims["N"] == int(cfg["sim"]["Tp"] / cfg["sim"]["Ts_MPC"]) == 38
    assert (dims["nh"], dims["nh_0"], dims["nh_e"]) == (1, 0, 1)                     # no gg row at stage 0
    assert (dims["ns_0"], dims["ns"], dims["ns_e"]) == (1, 3, 2)                     # penalty classes of cost_set
    assert (dims["nbx_0"], dims["nbx"], dims["nbu"], dims["nbx_e"]) == (dims["nx"], 1, 1, 1)
    assert (dims["ny"], dims["ny_e"]) == (6, 4)
    Q = np.diag([m["q_lon"] / m["s_lon"] ** 2, m["q_lat"] / m["s_lat"] ** 2, m["q_yaw"] / m["s_yaw"] ** 2, m["q_vel"] / m["s_vel"] ** 2])
    R = np.diag([m["r_jerk"] / m["s_jerk"] ** 2, m["r_steering_rate"] / m["s_steering_rate"] ** 2])
    W = np.zeros((6, 6)); W[:4, :4] = Q; W[4:, 4:] = R
    np.testing.assert_allclose(g["W"], 0.01 * W, rtol=1e-12)
    np.testing.assert_allclose(g["W_e"], 0.01 * Q, rtol=1e-12)
    for f in ("Zl", "Zu", "Zl_0", "Zl_e"):
        if f in g.files:
            assert np.all(g[f] == m["L2_pen"])
    for f in ("zl", "zu", "zl_0", "zl_e"):
        if f in g.files:
            assert np.all(g[f] == m["L1_pen"])
    assert g["lbx"][0] == veh["delta_f_min"] == g["lbx_e"][0] and g["ubx"][0] == veh["delta_f_max"] == g["ubx_e"][0]
    assert g["lbu"][0] == veh["delta_f_dot_min"] and g["ubu"][0] == veh["delta_f_dot_max"]
    assert (g["lh"][0], g["uh"][0], g["lh_e"][0], g["uh_e"][0]) == (0.0, 1.0, 0.0, 1.0)
    assert (int(g["idxbx"][0]), int(g["idxbu"][0]), int(g["idxbx_e"][0])) == (6, 1, 6)   # steering angle of the NOMINAL copy, steering rate
    assert str(g["cost_type"]) == str(g["cost_type_e"]) == "NONLINEAR_LS"
    assert opts["integrator_type"] == "DISCRETE" and opts["nlp_solver_type"] == "SQP_RTI"
    assert opts["qp_solver"] == "FULL_CONDENSING_HPIPM" and opts["hessian_approx"] == "GAUSS_NEWTON"
    assert opts["nlp_solver_step_length"] == 1.0 and opts["levenberg_marquardt"] == 0.0 and opts["regularize_method"] == "NO_REGULARIZE"
    d = make_desc(dims["N"], opts["tf"] / dims["N"], 1, 1, cfg=cfg)
    assert d.qp_iter_max == opts["qp_solver_iter_max"] == 50 and abs(d.dt - 0.08) < 1e-15 and d.nsub == 1


def test_oracle_model_functions_against_exported_expressions(golden_dir):
    """The oracle's stacked dynamics, cost output and chance constraint against the reference's OWN expressions: the exported
    OCP (acados_ocp_SNMPC.json) prints them as CasADi text, tests/golden/make_golden.py evaluated that text at 48 random
    points (both stop_flag values, the low-speed branch of the slip angles, the sloped part of the gg table). The printed
    constants carry 6 significant digits, hence the tolerance."""
    g = np.load(os.path.join(golden_dir, "snmpc_expr.npz"))
    n = g["X"].shape[0]
    worst = np.zeros(3)
    for j in range(n):
        o = orc.OracleSnmpcOcp(N=2, dt=float(g["Ts"]), Apce=g["A"][j], uph=1, gamma=float(g["gamma"]))
        xn, y, h = o.eval_stage(g["X"][j], g["U"][j], g["stop"][j])
        scale = 1.0 + np.abs(g["F"][j])
        worst[0] = max(worst[0], np.max(np.abs(xn - g["F"][j]) / scale))
        worst[1] = max(worst[1], np.max(np.abs(y - g["Y"][j]) / (1.0 + np.abs(g["Y"][j]))))
        worst[2] = max(worst[2], abs(h - g["H"][j]) / (1.0 + abs(g["H"][j])))
        np.testing.assert_allclose(y[:4], g["Ye"][j], rtol=0, atol=1e-4)
    assert worst[0] < 2e-5 and worst[1] < 2e-5 and worst[2] < 2e-5, worst
    assert set(g["stop"]) == {0.0, 1.0}


# ---------------------------------------------------------------------------------------------- GPU
def _gpu_vs_oracle(golden_dir, N, uph, poses, nsolve=3, shift_ref=0, kernel=None, prologue=None):
    from tum_control_amd.solver import CoupledSnmpcSolver
    snm, stds, w, A = _pce()
    d = np.load(os.path.join(golden_dir, "kat0.npz"))
    B = len(poses)
    import contextlib
    from tum_control_amd import solver as _sv
    with (_sv.dev_library() if kernel == "fused" else contextlib.nullcontext()):      # (the fused kernel: development build)
        s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=0.8, qp_warm_start=(False if kernel == "fused" else None))
    if kernel:
        s.set_kernel(kernel)
    if prologue:
        s.set_kernel(prologue)
    s.install_reference_ocp()
    X0 = np.zeros((B, 11, 8)); Y = np.zeros((B, N + 1, 6))
    rng = np.random.default_rng(11)
    for j, i in enumerate(poses):
        x0 = d["x0"][i].copy()
        if j > 0:
            x0[3:8] += rng.normal(0, 1, 5) * np.array([.8, .2, .04, .01, .3])
        X0[j] = snm.compute_x0dist(x0, w, stds)
        yr = d["yref"][i]
        Y[j, :min(N, 38) + 1, :4] = yr[:min(N, 38) + 1]
        for k in range(39, N + 1):                               # N = 40: extend the 39 logged reference points
            Y[j, k, :4] = 2 * Y[j, k - 1, :4] - Y[j, k - 2, :4]
    s.constraints_set(0, "lbx", X0.reshape(B, -1)); s.constraints_set(0, "ubx", X0.reshape(B, -1))
    s.set_yref_all(Y); s.cold_start()
    orcs = []
    from tum_control_amd import config
    m = config.MPC
    for j in range(B):
        o = orc.OracleSnmpcOcp(N=N, dt=0.08, Apce=A, uph=uph)
        o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
        o.yref[:] = Y[j]; o.cold_start(X0[j])
        if kernel == "fused":
            o.qp_warm_start(False)          # (the development build's kernels always cold-start the interior point method)
        orcs.append(o)
    for it in range(nsolve):
        assert s.solve() == 0
        Xn, U = s.get_iterate()
        cost = np.atleast_1d(s.get_cost())
        # Propagating the samples over more than ~30 stages makes the real-time iteration itself ill-conditioned: in the oracle
        # alone a 1e-9 perturbation of the iterate moves the NEXT solve by 2e-7 (uph = 33) / 2e-6 (uph = 38) against 1e-8 at
        # uph = 5 (measured, HISTORY.md (round-4 document, section 2)). The oracle therefore starts each warm iteration of a long propagation horizon
        # from the GPU's iterate (below), so that nothing accumulates from the solves before -- and even on IDENTICAL inputs the
        # two implementations of a warm solve at uph = 38 end 1.9e-6 apart (measured on the box: the condensed QP of that
        # iterate is conditioned badly enough that two interior point runs that both stop at 1e-8 differ at that level). The
        # cold start is held to 1e-7 for every uph, the re-seeded warm iterations of uph > 31 to 2e-5 relative / 5e-6 absolute
        # (2.5 times the measured deviation), those of uph > 40 -- the whole 48-stage horizon propagated: 1.7e-5 measured between
        # the column-per-lane prologue and the oracle, which sum the samples in a different order -- to 4e-5 absolute: all well
        # inside north_star's 1e-4.
        rt, at, rc = (1e-7, 2e-8, 1e-7) if (uph <= 31 or it == 0) else ((2e-5, 5e-6, 1e-6) if uph <= 40 else (4e-5, 4e-5, 2e-6))
        for j, o in enumerate(orcs):
            assert o.solve() == 0
            np.testing.assert_allclose(U[j], o.U, rtol=rt, atol=at, err_msg=f"U solve {it} inst {j}")
            np.testing.assert_allclose(Xn[j], o.X[:, 0], rtol=rt, atol=at, err_msg=f"X nominal solve {it} inst {j}")
            np.testing.assert_allclose(cost[j], o.cost, rtol=rc)
            for k in (0, 1, max(uph, 1), N):
                xf = np.atleast_2d(s.get(k, "x"))[j].reshape(11, 8)
                np.testing.assert_allclose(xf, o.X[k], rtol=rt, atol=at, err_msg=f"stacked x stage {k} solve {it} inst {j}")
        if uph > 31 and it + 1 < nsolve:
            XS = np.stack([np.atleast_2d(s.get(k, "x")).reshape(B, 11, 8) for k in range(N + 1)], axis=1)      # (B, N+1, 11, 8)
            for j, o in enumerate(orcs):
                assert np.abs(XS[j] - o.X).max() < 1e-5 * (1.0 + np.abs(o.X).max())
                o.X[:] = XS[j]; o.U[:] = U[j]
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("N,uph", [(38, 5), (38, 15), (40, 5), (38, 0), (40, 1), (12, 12)])
def test_gpu_coupled_snmpc_vs_oracle(golden_dir, N, uph):
    """cold start + two warm real-time iterations on logged poses (one with a perturbed state), every copy of the stacked
    iterate compared; tolerance 1e-7 relative (north_star: 1e-4)."""
    _gpu_vs_oracle(golden_dir, N, uph, poses=[0, 26, 30], kernel="fused")


@pytest.mark.gpu
@pytest.mark.parametrize("N,uph", [(38, 5), (38, 15), (40, 5), (38, 0), (12, 12), (44, 5), (48, 12), (40, 24), (40, 33), (38, 38), (48, 48), (50, 5), (56, 20), (52, 40)])
def test_gpu_coupled_snmpc_pipeline_vs_oracle(golden_dir, N, uph):
    """the same through the pipeline variant (prologue, lin_kernel<SN>, cond_kernel<., SN>, ipm_kernel, expand_kernel<., SN>,
    epilogue; what batches above 1024 instances run), including horizons beyond 40 (six-tile instantiation) and uncertainty
    propagation horizons up to the whole horizon (the reference ran UPH = Tp = 38 stages, SNMPC_class.py:103-104: the sample
    columns then no longer fit one wavefront and the prologue's hand-over buffer switches to the 128-column pitch). The
    propagation horizons also walk through the prologue's instantiations: column state in LDS (uph <= 8 at ten samples) and in
    registers for 6 / 9 / 13 / 17 passes (uph 15 / 24 / 33, 38 / 48)"""
    _gpu_vs_oracle(golden_dir, N, uph, poses=[0, 26, 30], kernel="pipeline")


@pytest.mark.gpu
@pytest.mark.parametrize("N,uph", [(38, 5), (38, 15), (40, 24), (40, 33), (38, 38), (48, 48)])
def test_gpu_coupled_snmpc_pipeline_passes_prologue_vs_oracle(golden_dir, N, uph):
    """the pipeline with the prologue of rounds 1-3 (set_kernel("prologue-passes"): column slots and passes; column state in LDS
    at uph = 5, in registers for 6 / 9 / 13 / 17 passes beyond) instead of the matrix-core prologue that is the default at
    ten samples: the second implementation of the same hand-over, held to the oracle like the first"""
    _gpu_vs_oracle(golden_dir, N, uph, poses=[0, 26, 30], kernel="pipeline", prologue="prologue-passes")


@pytest.mark.gpu
@pytest.mark.parametrize("N,uph", [(38, 5), (38, 15), (12, 12), (40, 1), (44, 5), (40, 24), (40, 33), (38, 38), (48, 48), (48, 32)])
def test_gpu_coupled_snmpc_pipeline_mfma_prologue_vs_oracle(golden_dir, N, uph):
    """the matrix-core prologue (column recursions as v_mfma_f64_4x4x4_4b products over the live column groups), named
    explicitly (it is also the library's own choice at ten samples, i.e. what the pipeline test above runs) on every
    propagation horizon: one and several column groups, the second phase for the columns beyond 63 (uph 32 is its first
    stage, 33 / 38 / 48 run it for 2 / 7 / 17 stages)"""
    _gpu_vs_oracle(golden_dir, N, uph, poses=[0, 26, 30], kernel="pipeline", prologue="prologue-mfma")


@pytest.mark.gpu
@pytest.mark.parametrize("N,uph,lib", [(12, 5, "shipped"), (12, 12, "shipped"), (40, 9, "shipped"), (40, 15, "shipped"), (40, 24, "shipped"), (40, 31, "shipped"), (40, 36, "shipped"),
                                       (38, 38, "shipped"), (12, 5, "dev"), (40, 9, "dev"), (40, 31, "dev")])
def test_gpu_coupled_snmpc_condensed_qp(golden_dir, N, uph, lib):
    """The condensed QP (H, q, chance / gg rows, constants) against the oracle's dense 88-state condensing, at a strongly
    excited iterate with non-zero defects on every copy: what the prologue + condensing kernel hand to the interior point
    kernel (shipped library: the dump is read back from the pipeline's hand-over buffers), and what the prologue + fused
    kernel build (development build)."""
    import contextlib
    from tum_control_amd import solver as _sv
    from tum_control_amd.solver import CoupledSnmpcSolver
    from tum_control_amd import config
    snm, stds, w, A = _pce()
    x0, yref, p = _kat(golden_dir)
    x0 = x0.copy(); x0[7] = 0.8; x0[5] = 0.12; x0[4] = 0.3
    rng = np.random.default_rng(5)
    U = np.stack([rng.normal(0, 1.0, N), rng.normal(0, 0.05, N)], axis=1)
    xs = snm.compute_x0dist(x0, w, stds)
    X, _ = _rollout(xs, U, A, uph, N, 0.08, 0.5)
    X += rng.normal(0, 1e-3, X.shape)                              # defects everywhere
    Y = np.zeros((N + 1, 6)); Y[:min(N, 38) + 1, :4] = yref[:min(N, 38) + 1]
    for k in range(39, N + 1):
        Y[k, :4] = 2 * Y[k - 1, :4] - Y[k - 2, :4]
    m = config.MPC
    o = orc.OracleSnmpcOcp(N=N, dt=0.08, Apce=A, uph=uph)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    o.yref[:] = Y; o.x0[:] = xs + 1e-3; o.X[:] = X; o.U[:] = U
    _, qp = o.solve_debug()
    with (_sv.dev_library() if lib == "dev" else contextlib.nullcontext()):
        s = CoupledSnmpcSolver(N=N, dt=0.08, batch=1, Apce=A, uph=uph, gamma=0.8, qp_warm_start=(False if lib == "dev" else None))
    s.install_reference_ocp()
    s.constraints_set(0, "lbx", (xs + 1e-3).flatten()); s.constraints_set(0, "ubx", (xs + 1e-3).flatten())
    s.set_yref_all(Y)
    for k in range(N + 1):
        s.set(k, "x", X[k].flatten())
    s.set_iterate(U=U)
    dbg = s.debug_dump(0)
    nv = 2 * N
    H = dbg[:6400].reshape(80, 80)[:nv, :nv]
    q = dbg[6400:6400 + nv]
    rows = dbg[6480:6480 + 2 * N * 80].reshape(2 * N, 80)[:, :nv]
    dd = dbg[12880:12880 + 2 * N]
    sc = np.abs(qp["H"]).max()
    np.testing.assert_allclose(H, qp["H"], rtol=0, atol=1e-11 * sc)
    np.testing.assert_allclose(q, qp["q"], rtol=0, atol=1e-11 * np.abs(qp["q"]).max())
    Ch = qp["C"][N + 1::2]
    np.testing.assert_allclose(rows[1::2], Ch, rtol=0, atol=1e-11 * max(1.0, np.abs(Ch).max()))
    np.testing.assert_allclose(rows[0::2], qp["C"][N::2], rtol=0, atol=1e-13)
    np.testing.assert_allclose(dd[1::2], qp["d"][N + 1::2], rtol=0, atol=1e-11)
    np.testing.assert_allclose(dd[0::2], qp["d"][N::2], rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("ns,L,N,uph,prologue", [(15, 10, 20, 6, None), (16, 16, 14, 14, None), (7, 4, 40, 11, None), (1, 1, 10, 3, None), (3, 2, 40, 31, None),
                                                 (8, 6, 40, 12, "prologue-passes"), (8, 6, 40, 24, "prologue-passes"), (8, 6, 38, 38, "prologue-passes"),
                                                 (9, 6, 40, 24, "prologue-passes"), (7, 4, 40, 11, "prologue-passes"), (3, 2, 40, 31, "prologue-passes"),
                                                 (8, 6, 38, 38, None), (9, 6, 40, 24, None), (6, 3, 40, 36, None), (2, 2, 40, 40, None),
                                                                                                  (7, 4, 40, 11, "prologue-mfma"), (1, 1, 10, 3, "prologue-mfma"), (10, 10, 40, 9, "prologue-mfma"), (8, 6, 38, 38, "prologue-mfma"),
                                                 (9, 6, 40, 24, "prologue-mfma"), (6, 3, 40, 36, "prologue-mfma"), (2, 2, 40, 40, "prologue-mfma"), (3, 2, 40, 31, "prologue-mfma"),
                                                 (5, 4, 40, 17, "prologue-mfma")])
def test_gpu_coupled_snmpc_other_sample_counts(golden_dir, ns, L, N, uph, prologue):
    """sample counts / PCE sizes other than the shipped 10 x 10 (any L x n_s matrix defines a valid OCP): condensed QP and
    one full step of every copy against the oracle. Up to ten samples run the matrix-core prologue (five samples per
    wavefront; fewer than six leave the second wavefront of the workgroup without a sample), more than ten -- or
    set_kernel("prologue-passes") -- the column-slot / pass kernels. Of those, eight samples with a propagation horizon of
    12 / 24 / 38 stages run the register-resident instantiations (6 / 9 / 13 passes) with EIGHT column slots per sample: the 64
    lanes then only reach eight of the nine reduction rows, and the chance-constraint row takes a second round (round 3 dropped
    it)."""
    from tum_control_amd.solver import CoupledSnmpcSolver
    from tum_control_amd import config
    x0, yref, p = _kat(golden_dir)
    x0 = x0.copy(); x0[7] = -0.6; x0[5] = 0.1; x0[4] = -0.2
    rng = np.random.default_rng(100 + ns)
    A = rng.normal(0, 0.3, (L, ns)); A[0] = np.abs(A[0]) + 0.1; A[0] /= A[0].sum()
    xs = np.tile(x0, (ns + 1, 1)); xs[1:, 3:6] += rng.normal(0, 1, (ns, 3)) * np.array([.8, .35, .035])
    U = np.stack([rng.normal(0, 1.0, N), rng.normal(0, 0.05, N)], axis=1)
    X, _ = _rollout(xs, U, A, uph, N, 0.08, 0.5)
    X += rng.normal(0, 1e-3, X.shape)
    Y = np.zeros((N + 1, 6)); Y[:min(N, 38) + 1, :4] = yref[:min(N, 38) + 1]
    for k in range(39, N + 1):
        Y[k, :4] = 2 * Y[k - 1, :4] - Y[k - 2, :4]
    m = config.MPC
    o = orc.OracleSnmpcOcp(N=N, dt=0.08, Apce=A, uph=uph)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    o.yref[:] = Y; o.x0[:] = xs; o.X[:] = X; o.U[:] = U
    s = CoupledSnmpcSolver(N=N, dt=0.08, batch=1, Apce=A, uph=uph, gamma=0.8)
    if prologue:
        s.set_kernel(prologue)
    s.install_reference_ocp()
    s.constraints_set(0, "lbx", xs.flatten()); s.constraints_set(0, "ubx", xs.flatten())
    s.set_yref_all(Y)
    for k in range(N + 1):
        s.set(k, "x", X[k].flatten())
    s.set_iterate(U=U)
    dbg = s.debug_dump(0)                                          # (this is a solve)
    _, qp = o.solve_debug()
    nv = 2 * N
    H = dbg[:6400].reshape(80, 80)[:nv, :nv]
    rows = dbg[6480:6480 + 2 * N * 80].reshape(2 * N, 80)[:, :nv]
    np.testing.assert_allclose(H, qp["H"], rtol=0, atol=1e-11 * np.abs(qp["H"]).max())
    Ch = qp["C"][N + 1::2]
    np.testing.assert_allclose(rows[1::2], Ch, rtol=0, atol=1e-11 * max(1.0, np.abs(Ch).max()))
    np.testing.assert_allclose(dbg[12880:12880 + 2 * N][1::2], qp["d"][N + 1::2], rtol=0, atol=1e-11)
    assert o.status == 0 and int(np.atleast_1d(s.get_stats("status"))[0]) == 0
    Xn, Un = s.get_iterate()
    np.testing.assert_allclose(Un[0], o.U, rtol=1e-6, atol=1e-7)
    for k in (0, 1, uph, N):
        np.testing.assert_allclose(s.get(k, "x").reshape(ns + 1, 8), o.X[k], rtol=1e-6, atol=1e-7, err_msg=f"stage {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("ns,L,N,uph", [(10, 10, 38, 5), (15, 10, 20, 6), (17, 17, 40, 3), (20, 20, 38, 5), (24, 10, 20, 8), (32, 32, 12, 6), (12, 20, 38, 5), (20, 20, 50, 4), (24, 20, 38, 18), (32, 20, 40, 16),
                                        (17, 5, 44, 28)])
def test_gpu_coupled_snmpc_more_than_sixteen_samples(golden_dir, ns, L, N, uph):
    """n_samples and the number of PCE terms up to 32 (round 6; 16 before: MPC_params.yaml's n_samples / expansion_degree are free parameters of the
    reference, SNMPC_class.py:78-94 -- degree 3 in three variables is 20 terms). Beyond 16 of either the capsule runs the column-slot prologue with its
    column state in LDS and the epilogue in their 32-wide instantiations; everything up to 16 runs the kernels of rounds 1-5 unchanged. One solve from a
    perturbed iterate against the oracle: inputs, nominal copy and the stacked sample copies of a few stages."""
    from tum_control_amd.solver import CoupledSnmpcSolver
    from tum_control_amd import config
    x0, yref, p = _kat(golden_dir)
    x0 = x0.copy(); x0[7] = -0.6; x0[5] = 0.1; x0[4] = -0.2
    rng = np.random.default_rng(100 + ns + L)
    A = rng.normal(0, 0.3 / np.sqrt(ns / 10.0), (L, ns)); A[0] = np.abs(A[0]) + 0.1; A[0] /= A[0].sum()
    xs = np.tile(x0, (ns + 1, 1)); xs[1:, 3:6] += rng.normal(0, 1, (ns, 3)) * np.array([.8, .35, .035])
    U = np.stack([rng.normal(0, 1.0, N), rng.normal(0, 0.05, N)], axis=1)
    X, _ = _rollout(xs, U, A, uph, N, 0.08, 0.5)
    X += rng.normal(0, 1e-3, X.shape)
    Y = np.zeros((N + 1, 6)); Y[:min(N, 38) + 1, :4] = yref[:min(N, 38) + 1]
    for k in range(39, N + 1):
        Y[k, :4] = 2 * Y[k - 1, :4] - Y[k - 2, :4]
    m = config.MPC
    o = orc.OracleSnmpcOcp(N=N, dt=0.08, Apce=A, uph=uph)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    o.yref[:] = Y; o.x0[:] = xs; o.X[:] = X; o.U[:] = U
    s = CoupledSnmpcSolver(N=N, dt=0.08, batch=1, Apce=A, uph=uph, gamma=0.8)
    s.install_reference_ocp()
    s.constraints_set(0, "lbx", xs.flatten()); s.constraints_set(0, "ubx", xs.flatten())
    s.set_yref_all(Y)
    for k in range(N + 1):
        s.set(k, "x", X[k].flatten())
    s.set_iterate(U=U)
    assert s.solve() == 0 and o.solve() == 0
    Xn, Un = s.get_iterate()
    np.testing.assert_allclose(Un.reshape(N, 2), o.U, rtol=1e-6, atol=1e-7)
    for k in (0, 1, uph, N):
        np.testing.assert_allclose(np.asarray(s.get(k, "x")).reshape(ns + 1, 8), o.X[k], rtol=1e-6, atol=1e-7, err_msg=f"stage {k}")
    assert abs(float(np.atleast_1d(s.get_cost())[0]) - o.cost) < 1e-7 * max(1.0, abs(o.cost))
    with pytest.raises(Exception, match="1..32"):
        CoupledSnmpcSolver(N=10, batch=1, Apce=np.zeros((3, 33)), uph=2)
    with pytest.raises(Exception, match="too large for the prologue kernel's LDS"):      # (128 KiB of column state: uph <= 28 / 26 / 18 / 16 at 17 / 20 / 24 / 32 samples)
        CoupledSnmpcSolver(N=40, batch=1, Apce=np.zeros((10, 24)), uph=19)


@pytest.mark.gpu
def test_gpu_snmpc_frozen_sample_copies_are_deferred_not_lost():
    """The epilogue does not write the sample copies of the stages > uph (they equal stage uph, pred_model_dynamic_disc.py:203);
    they are brought up to date when asked for. Two solves without a read in between, then reads / a write beyond uph."""
    from tum_control_amd.solver import CoupledSnmpcSolver
    from tum_control_amd.workloads import nominal_batch
    snm, stds, w, A = _pce()
    N, uph, ns, B = 40, 5, 10, 6
    x0, yref = nominal_batch(B, N=N, seed=11)
    s = CoupledSnmpcSolver(N=N, batch=B, Apce=A, uph=uph, x0_offsets=snm.x0_offsets(w, stds))
    s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
    assert s.solve() == 0 and s.solve() == 0
    at = lambda k: np.atleast_2d(s.get(k, "x")).reshape(B, ns + 1, 8)
    xu = at(uph)
    for k in (uph + 1, 17, N):
        np.testing.assert_array_equal(at(k)[:, 1:], xu[:, 1:], err_msg=f"stage {k}")
    assert s.solve() == 0                       # a third solve, then a WRITE beyond uph before any read
    xu3 = at(uph)
    mine = np.arange(8.0 * (ns + 1)).reshape(1, -1) + 0.25
    s.set(N - 1, "x", np.repeat(mine, B, axis=0))
    np.testing.assert_array_equal(at(N - 1), np.repeat(mine, B, axis=0).reshape(B, ns + 1, 8))
    np.testing.assert_array_equal(at(N)[:, 1:], xu3[:, 1:])          # the other stages got the third solve's copies
    assert not np.array_equal(xu3[:, 1:], xu[:, 1:])


@pytest.mark.gpu
def test_gpu_snmpc_longer_horizon_after_deferred_freeze():
    """A warm solve with a LONGER uncertainty propagation horizon reads sample copies of stages the previous solves left
    to be frozen later: the change of uph must bring them up to date first. Reference: a second capsule, attached with the
    longer horizon, that is handed the first one's complete iterate (every stage read back, i.e. frozen copies included)."""
    from tum_control_amd.solver import CoupledSnmpcSolver
    from tum_control_amd.workloads import nominal_batch
    snm, stds, w, A = _pce()
    N, ns, B = 38, 10, 4
    x0, yref = nominal_batch(B, N=N, seed=21)
    off = snm.x0_offsets(w, stds)

    # (the second capsule is handed the ITERATE of the first, not the multipliers of its last QP: both cold-start the interior point method)
    def warm(uph0):
        s = CoupledSnmpcSolver(N=N, batch=B, Apce=A, uph=uph0, gamma=0.8, x0_offsets=off, qp_warm_start=False)
        s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
        assert s.solve() == 0 and s.solve() == 0
        return s

    def setp(s, uph):
        for k in range(N + 1):
            s.set(k, "p", np.concatenate([A.flatten(), [0.8], [1.0 if k >= uph else 0.0]]))

    a = warm(5); setp(a, 9)
    assert a.solve() == 0                                   # stages 6..9 of the samples: frozen copies of the uph = 5 solves
    b = warm(5)
    stacked = [np.atleast_2d(b.get(k, "x")).copy() for k in range(N + 1)]
    U = b.get_iterate()[1].copy()
    c = CoupledSnmpcSolver(N=N, batch=B, Apce=A, uph=9, gamma=0.8, x0_offsets=off, qp_warm_start=False)
    c.install_reference_ocp(); c.set_x0(x0); c.set_yref_all(yref); c.cold_start()
    for k in range(N + 1):
        c.set(k, "x", stacked[k])
    c.set_iterate(U=U)
    assert c.solve() == 0
    np.testing.assert_array_equal(a.get_iterate()[1], c.get_iterate()[1])
    for k in (0, 5, 9, N):
        np.testing.assert_array_equal(np.atleast_2d(a.get(k, "x")), np.atleast_2d(c.get(k, "x")))


@pytest.mark.gpu
def test_gpu_snmpc_controller_mirror(golden_dir):
    """the SNMPC_class.py mirror drives the coupled solver like the reference's controller does"""
    from tum_control_amd.snmpc import Stochastic_Nonlinear_Model_Predictive_Controller as C
    snm, stds, w, A = _pce()
    x0, yref, p = _kat(golden_dir)
    c = C(X0_MPC=x0)
    ref = dict(pos_x=yref[:, 0], pos_y=yref[:, 1], ref_yaw=yref[:, 2], ref_v=yref[:, 3])
    u0, pred, stats = c.solve(ref)
    from tum_control_amd import config
    m = config.MPC
    o = orc.OracleSnmpcOcp(N=c.N, dt=c.Tp / c.N, Apce=A, uph=5)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    o.yref[:, :4] = yref; o.cold_start(snm.compute_x0dist(x0, w, stds))
    assert o.solve() == 0 and stats[4] == 0
    np.testing.assert_allclose(u0, o.U[0], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(pred, o.X[:c.N, 0], rtol=1e-7, atol=2e-8)
    np.testing.assert_allclose(stats[0], o.cost, rtol=1e-7)
    # next control step: new initial state, warm start
    x1 = pred[1].copy()
    c.set_initial_state(x1)
    o.set_initial_state(snm.compute_x0dist(x1, w, stds))
    u0, pred, stats = c.solve(ref)
    assert o.solve() == 0
    np.testing.assert_allclose(u0, o.U[0], rtol=1e-7, atol=1e-9)
    # reset protocol (SNMPC_class.py:266-272)
    c.reset(x0)
    xf = c.acados_solver.get(3, "x")
    np.testing.assert_array_equal(xf, snm.compute_x0dist(x0, w, stds).flatten())


@pytest.mark.gpu
def test_gpu_snmpc_closed_loop(golden_dir):
    """The SNMPC controller in closed loop (planner -> coupled solve -> plant -> estimator, main.py:48-78 with
    MPC_type SNMPC): host loop around the GPU solver vs the same loop around the oracle, and the loop that never leaves
    the device (x0 fan-out to the samples as a kernel) vs the host loop."""
    from tum_control_amd.closed_loop import ClosedLoopBatch, plant_step, MovingAverageEstimator
    from tum_control_amd.planner import planner_emulator, yref_from_ref
    from tum_control_amd import config
    snm, stds, w, A = _pce()
    steps, N, Tp = 30, 38, 3.04
    cl = ClosedLoopBatch("monteblanco", batch=1, N=N, Tp=Tp, controller="snmpc")
    lg = cl.run(steps)
    # the same loop around the oracle
    m = config.MPC
    o = orc.OracleSnmpcOcp(N=N, dt=Tp / N, Apce=A, uph=5)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    x_mpc = lg["MPC_SimX"][0][0].copy(); x_sim = x_mpc[:7].copy()[None]; pose = x_mpc[:2].copy()
    o.cold_start(snm.compute_x0dist(x_mpc, w, stds))
    est = MovingAverageEstimator(1)
    cfg = config.default_config()
    for t in range(steps):
        _, ref = planner_emulator(cl.track, pose, N + 1, Tp, True)
        o.yref[:] = yref_from_ref(ref, N)
        assert o.solve() == 0
        u0, x1 = o.U[0].copy(), o.X[1, 0].copy()
        np.testing.assert_allclose(lg["simU"][t][0], u0, rtol=1e-6, atol=1e-7, err_msg=f"step {t}")
        x_sim = plant_step(x_sim, np.array([x1[7]]), np.array([u0[1]]), cfg, 0.02)
        pose = x_sim[0, :2].copy()
        x_mpc = est(np.concatenate([x_sim, [[x1[7]]]], axis=1))[0]
        o.set_initial_state(snm.compute_x0dist(x_mpc, w, stds))
    np.testing.assert_allclose(lg["CiLX"][-1][0], x_sim[0], rtol=1e-7, atol=1e-7)
    assert (lg["simSolverDebug"][:, 0, 4] == 0).all()
    # entirely on the device, three vehicles
    cd = ClosedLoopBatch("monteblanco", batch=3, N=N, Tp=Tp, controller="snmpc", on_device=True, log_capacity=steps)
    ld = cd.run(steps)
    for b in range(3):
        np.testing.assert_allclose(ld["simU"][:, b], lg["simU"][:, 0], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ld["CiLX"][:, b], lg["CiLX"][:, 0], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_gpu_snmpc_closed_loop_degree_three_expansion(golden_dir):
    """MPC_params.yaml's n_samples / expansion_degree beyond the shipped 10 / 2 (SNMPC_class.py:78-94: free parameters of the reference): 24 Hammersley
    samples and the degree-3 expansion in the three uncertain states (20 PCE terms), which the 32-wide instantiations of the prologue and the epilogue
    run (round 6). The closed loop around the GPU solver against the same loop around the oracle, and the on-device loop against the host loop."""
    import copy
    from tum_control_amd.closed_loop import ClosedLoopBatch, plant_step, MovingAverageEstimator
    from tum_control_amd.planner import planner_emulator, yref_from_ref
    from tum_control_amd import config, snmpc as snm
    cfg = copy.deepcopy(config.default_config())
    cfg["mpc"]["n_samples"], cfg["mpc"]["expansion_degree"], cfg["mpc"]["uncertainty_propagation_horizon"] = 24, 3, 7
    stds = np.asarray(cfg["mpc"]["stds"], dtype=float)
    w = snm.hammersley_normal(24, 3)
    A = snm.pce_matrix(w, snm.alpha_generation(3, 3))
    assert A.shape == (20, 24)
    steps, N, Tp = 24, 38, 3.04
    cl = ClosedLoopBatch("monteblanco", batch=1, N=N, Tp=Tp, controller="snmpc", cfg=cfg)
    lg = cl.run(steps)
    m = cfg["mpc"]
    o = orc.OracleSnmpcOcp(N=N, dt=Tp / N, Apce=A, uph=7)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    x_mpc = lg["MPC_SimX"][0][0].copy(); x_sim = x_mpc[:7].copy()[None]; pose = x_mpc[:2].copy()
    o.cold_start(snm.compute_x0dist(x_mpc, w, stds))
    est = MovingAverageEstimator(1)
    for t in range(steps):
        _, ref = planner_emulator(cl.track, pose, N + 1, Tp, True)
        o.yref[:] = yref_from_ref(ref, N)
        assert o.solve() == 0
        u0, x1 = o.U[0].copy(), o.X[1, 0].copy()
        np.testing.assert_allclose(lg["simU"][t][0], u0, rtol=1e-6, atol=1e-7, err_msg=f"step {t}")
        x_sim = plant_step(x_sim, np.array([x1[7]]), np.array([u0[1]]), cfg, 0.02)
        pose = x_sim[0, :2].copy()
        x_mpc = est(np.concatenate([x_sim, [[x1[7]]]], axis=1))[0]
        o.set_initial_state(snm.compute_x0dist(x_mpc, w, stds))
    np.testing.assert_allclose(lg["CiLX"][-1][0], x_sim[0], rtol=1e-7, atol=1e-7)
    assert (lg["simSolverDebug"][:, 0, 4] == 0).all()
    cd = ClosedLoopBatch("monteblanco", batch=3, N=N, Tp=Tp, controller="snmpc", cfg=cfg, on_device=True, log_capacity=steps)
    ld = cd.run(steps)
    for b in range(3):
        np.testing.assert_allclose(ld["simU"][:, b], lg["simU"][:, 0], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ld["CiLX"][:, b], lg["CiLX"][:, 0], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_gpu_snmpc_errors():
    """acados-style failures of the coupled solver: wrong dimensions and unsupported configurations raise, nothing is silent"""
    from tum_control_amd.solver import CoupledSnmpcSolver, BatchedOcpSolver, _dp
    snm, stds, w, A = _pce()
    with pytest.raises(Exception, match="propagation horizon"):
        CoupledSnmpcSolver(N=40, batch=1, Apce=A, uph=41)
    from tum_control_amd import solver as _sv
    with _sv.dev_library():
        f = CoupledSnmpcSolver(N=40, batch=1, Apce=A, uph=32, qp_warm_start=False)       # beyond 31 stages: pipeline only
    f.install_reference_ocp(); f.set_kernel("fused")
    with pytest.raises(Exception, match="fused"):
        f.solve()
    with pytest.raises(Exception, match="n_samples"):
        CoupledSnmpcSolver(N=10, batch=1, Apce=np.zeros((3, 33)), uph=2)
    n = BatchedOcpSolver(N=10, nsub=3, batch=1)
    assert n._L.tum_ocp_snmpc_attach(n._h, 10, 10, _dp(A), 5, 0.8) != 0 and "nsub = 1" in n._err()
    s = CoupledSnmpcSolver(N=10, batch=2, Apce=A, uph=3)
    with pytest.raises(Exception, match="snmpc_set_offsets"):
        s.set_x0(np.zeros(8))                                   # 8 values need the registered sample offsets
    with pytest.raises(Exception, match="mismatching dimension"):
        s.set(1, "x", np.zeros(87))
    with pytest.raises(Exception):
        s.get_from_qp_in(0, "A")
    s.set_x0_offsets(snm.x0_offsets(w, stds))
    x0 = np.array([[0, 0, 0.3, 30, 0, 0, 0, 0.0], [1, 2, 0.1, 25, 0.1, 0, 0, 0.5]])
    s.set_x0(x0); s.cold_start()
    full = s.get(4, "x").reshape(2, 11, 8)                      # fan-out happened on the device
    np.testing.assert_array_equal(full[:, 0], x0)
    np.testing.assert_allclose(full[:, 1:], x0[:, None, :] + snm.x0_offsets(w, stds)[None], rtol=0, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nominal", "snmpc", "r2"])
def test_gpu_main_loop_protocol(golden_dir, kind):
    """The reference's main.py:38-74 drives any of its three controller classes through the same few calls; the mirrors
    answer all of them (MPC.nx, MPC.model.u.size(), solve, reintialize_solver on failure, set_initial_state)."""
    from tum_control_amd.nmpc import Nonlinear_Model_Predictive_Controller as Nominal
    from tum_control_amd.snmpc import Stochastic_Nonlinear_Model_Predictive_Controller as Stochastic
    from tum_control_amd.r2nmpc import Reduced_Robustified_Nonlinear_Model_Predictive_Controller as Robust
    from tum_control_amd.closed_loop import plant_step, MovingAverageEstimator
    from tum_control_amd.planner import load_track, planner_emulator
    from tum_control_amd import config
    cfg = config.default_config()
    x0, _, _ = _kat(golden_dir)
    MPC = {"nominal": Nominal, "snmpc": Stochastic, "r2": Robust}[kind](None, None, None, x0)
    nx, nu = MPC.nx, MPC.model.u.size()[0]
    assert (nx, nu) == (8, 2)
    track = load_track("monteblanco")
    x_sim = x0[:7].copy()[None]; pose = x0[:2].copy(); est = MovingAverageEstimator(1)
    for i in range(6):
        _, ref = planner_emulator(track, pose, MPC.N + 1, MPC.Tp, True)
        traj = dict(pos_x=ref[:, 0], pos_y=ref[:, 1], ref_yaw=ref[:, 2], ref_v=ref[:, 3])
        u0, pred_X, stats = MPC.solve(traj)
        assert stats[-1] == 0 and pred_X.shape == (MPC.N, nx) and np.isfinite(u0).all()
        x_sim = plant_step(x_sim, np.array([pred_X[1, 7]]), np.array([u0[1]]), cfg, 0.02)
        pose = x_sim[0, :2].copy()
        x_next = est(np.concatenate([x_sim, [[pred_X[1, 7]]]], axis=1))[0]
        if i == 3:
            MPC.reintialize_solver(x_next)          # what main.py does after a failed solve
        MPC.set_initial_state(x_next)
    assert abs(x_sim[0, 4]) < 0.5                   # still driving straight down the start straight


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_gpu_coupled_snmpc_randomised(golden_dir, seed):
    """random horizons, propagation horizons, sample counts, PCE matrices, weights, tightened bounds and iterates: one
    real-time iteration of every instance of a small batch against the oracle (instances whose QP hit the iteration cap
    in either implementation are compared at the looser 1e-4 of north_star)."""
    from tum_control_amd.solver import CoupledSnmpcSolver, _dp
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(5, 41)); uph = int(rng.integers(0, min(N, 31) + 1)); ns = int(rng.integers(1, 17)); L = int(rng.integers(1, 17))
    B = 5
    d = np.load(os.path.join(golden_dir, "kat0.npz"))
    A = rng.normal(0, 0.25, (L, ns)); A[0] = np.abs(A[0]) + 0.05; A[0] /= A[0].sum()
    s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=float(rng.uniform(0.6, 0.95)))
    s.install_reference_ocp()
    q = np.array([2.8, 0.4, 0.2, 38.1, 101.4]) * 0.01 * rng.uniform(0.3, 3.0, 5)
    W = np.diag([q[0], q[0], q[1], q[2], q[3], q[4]])
    s.cost_set(-1, "W", W); s.cost_set(N, "W", W[:4, :4])
    L1, L2 = float(rng.uniform(20, 200)), float(rng.uniform(2, 50))
    for st, n in ((0, 1), (1, 3), (N, 2)) if N > 1 else ((0, 1), (N, 2)):
        for f, v in (("zl", L1), ("zu", L1), ("Zl", L2), ("Zu", L2)):
            s.cost_set(st, f, np.ones(n) * v)
    uh = float(rng.uniform(0.05, 1.0)); sr = float(rng.uniform(0.02, 0.322))
    for k in range(1, N + 1):
        s.constraints_set(k, "uh", np.array([uh]))
    for k in range(N):
        s.constraints_set(k, "lbu", np.array([-sr])); s.constraints_set(k, "ubu", np.array([sr]))
    orcs = []
    for j in range(B):
        i = int(rng.integers(0, 52))
        x0 = d["x0"][i].copy(); x0[3:8] += rng.normal(0, 1, 5) * np.array([1.5, .3, .05, .02, .5])
        xs = np.tile(x0, (ns + 1, 1)); xs[1:, 3:6] += rng.normal(0, 1, (ns, 3)) * np.array([.8, .35, .035])
        U = np.stack([rng.normal(0, 0.5, N), rng.normal(0, 0.03, N)], axis=1)
        X, _ = _rollout(xs, U, A, uph, N, 0.08, 0.5)
        X += rng.normal(0, 2e-3, X.shape)
        Y = np.zeros((N + 1, 6)); Y[:min(N, 38) + 1, :4] = d["yref"][i][:min(N, 38) + 1]
        for k in range(39, N + 1):
            Y[k, :4] = 2 * Y[k - 1, :4] - Y[k - 2, :4]
        o = orc.OracleSnmpcOcp(N=N, dt=0.08, Apce=A, uph=uph, gamma=0.8)
        o._view("kappa")[0] = np.sqrt((1 - s.gamma) / s.gamma)
        o.W[:] = np.diag(W); o.zl[:] = L1; o.zu[:] = L1; o.Zl[:] = L2; o.Zu[:] = L2
        o.uh[:] = uh; o.lbu[:] = -sr; o.ubu[:] = sr
        o.yref[:] = Y; o.x0[:] = xs; o.X[:] = X; o.U[:] = U
        orcs.append(o)
        s._chk(s._L.tum_ocp_constraints_set(s._h, 0, b"lbx", _dp(xs),
                                            xs.size, j, 1, xs.size), "lbx")
        for k in range(N + 1):
            v = np.ascontiguousarray(X[k].reshape(-1))
            s._chk(s._L.tum_ocp_set(s._h, k, b"x", _dp(v), v.size, j, 1, v.size), "x")
        s._chk(s._L.tum_ocp_set(s._h, -1, b"u", _dp(np.ascontiguousarray(U.reshape(-1))),
                                2 * N, j, 1, 2 * N), "u")
        s._chk(s._L.tum_ocp_set(s._h, -1, b"yref", _dp(np.ascontiguousarray(Y.reshape(-1))),
                                6 * (N + 1), j, 1, 6 * (N + 1)), "yref")
    s.solve()
    Xn, Un = s.get_iterate()
    it = s.get_stats("qp_iter"); st = s.get_stats("status")
    for j, o in enumerate(orcs):
        so = o.solve()
        assert so == int(st[j]), f"status inst {j}"
        if so != 0:
            continue
        tol = 1e-6 if (it[j] < 50 and o.qp_iter < 50) else 1e-4
        np.testing.assert_allclose(Un[j], o.U, rtol=tol, atol=tol, err_msg=f"seed {seed} N {N} uph {uph} ns {ns} L {L} inst {j} it {it[j]}/{o.qp_iter}")
        np.testing.assert_allclose(Xn[j], o.X[:, 0], rtol=tol, atol=tol)
        full = np.zeros((B, 8 * (ns + 1)))
        s._chk(s._L.tum_ocp_get(s._h, min(uph, N), b"x", _dp(full),
                                8 * (ns + 1), 0, B, 8 * (ns + 1)), "get")
        np.testing.assert_allclose(full[j].reshape(ns + 1, 8), o.X[min(uph, N)], rtol=tol, atol=tol)


@pytest.mark.gpu
def test_gpu_installed_bounds_match_exported_ocp(golden_dir):
    """what install_reference_ocp leaves in the solver, read back through the C-ABI, against acados_ocp_SNMPC.json"""
    from tum_control_amd.solver import CoupledSnmpcSolver
    g = np.load(os.path.join(golden_dir, "snmpc_json.npz"))
    snm, stds, w, A = _pce()
    s = CoupledSnmpcSolver(N=38, dt=0.08, batch=1, Apce=A, uph=5)
    s.install_reference_ocp()
    for k in (1, 17, 37):
        assert float(np.atleast_1d(s.constraints_get(k, "lbx"))[0]) == g["lbx"][0] and float(np.atleast_1d(s.constraints_get(k, "ubx"))[0]) == g["ubx"][0]
        assert float(np.atleast_1d(s.constraints_get(k, "lh"))[0]) == g["lh"][0] and float(np.atleast_1d(s.constraints_get(k, "uh"))[0]) == g["uh"][0]
    for k in (0, 20, 37):
        assert float(np.atleast_1d(s.constraints_get(k, "lbu"))[0]) == g["lbu"][0] and float(np.atleast_1d(s.constraints_get(k, "ubu"))[0]) == g["ubu"][0]
    assert float(np.atleast_1d(s.constraints_get(38, "ubx"))[0]) == g["ubx_e"][0] and float(np.atleast_1d(s.constraints_get(38, "uh"))[0]) == g["uh_e"][0]
    assert s.nx == 88 and s.L * s.ns + 2 == int(g["n_param"])


@pytest.mark.gpu
def test_gpu_snmpc_nan_isolation(golden_dir):
    """a poisoned instance fails alone (status 4, iterate untouched) and its neighbours are bit-identical to a clean batch"""
    from tum_control_amd.solver import CoupledSnmpcSolver
    snm, stds, w, A = _pce()
    x0, yref, p = _kat(golden_dir)
    N, B = 38, 3
    Y = np.zeros((N + 1, 6)); Y[:, :4] = yref
    xs = snm.compute_x0dist(x0, w, stds)

    def run(poison):
        s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=5)
        s.install_reference_ocp()
        X0 = np.tile(xs.reshape(1, -1), (B, 1))
        if poison:
            X0[1, 8 * 3 + 4] = np.nan                     # lateral speed of sample 3 of instance 1
        s.constraints_set(0, "lbx", X0); s.constraints_set(0, "ubx", X0)
        s.set_yref_all(Y); s.cold_start()
        st = s.solve()
        Xn, U = s.get_iterate()
        return st, s.get_stats("status").copy(), Xn, U, np.array([np.atleast_2d(s.get(2, "x"))[j] for j in range(B)])

    st0, stat0, X0n, U0, F0 = run(False)
    st1, stat1, X1n, U1, F1 = run(True)
    assert st0 == 0 and list(stat0) == [0, 0, 0]
    assert st1 == 4 and list(stat1) == [0, 4, 0]
    for j in (0, 2):
        np.testing.assert_array_equal(U1[j], U0[j]); np.testing.assert_array_equal(X1n[j], X0n[j]); np.testing.assert_array_equal(F1[j], F0[j])
    assert np.all(U1[1] == 0.0)                           # failed instance: inputs stay at the cold start


def test_casadi_text_evaluator():
    """the evaluator that turned the exported CasADi text into snmpc_expr.npz, on hand-checkable expressions of the same
    printed form (sub-expression definitions, selections, function calls with output selectors, element assignment,
    column-major reshape / slice / transpose, mac)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import casadi_expr as ce
    prog = ("@1=(2<a), @2=sqrt((sq(a)+sq(b))), @3=vertcat(a, b, ((@1?(6.28319+@2):0)+((!@1)?@2:0))), "
            "@4=g((@3+((0.08*@3)/2)), 7.35294e-05){0}, "
            "vertcat(@4, mac(reshape(M)',(dense((project((zeros(2x1,1nz)[0] = (a/2.)))[1] = (-b)))),zeros(2x1))[:2:1]', fmax(fmin((-(a/4)),0.98),-0.98))")
    defs, res = ce.parse_program(prog)
    M = np.array([1.0, 2.0, 3.0, 4.0])                                    # reshape(M, 2, 2) column-major = [[1,3],[2,4]]
    for a, b in ((3.0, 4.0), (1.0, -2.0)):
        ev = ce.Evaluator(defs, dict(a=a, b=b, M=M), {"g": lambda e, x, s: np.asarray(x) * 2.0 + s},
                          lambda ast, v: np.asarray(v, float).reshape((2, 2), order="F"))
        out = ev.ev(res)
        n = np.hypot(a, b)
        t3 = np.array([a, b, 6.28319 + n if 2 < a else n])
        g = (t3 + 0.08 * t3 / 2) * 2.0 + 7.35294e-05
        mv = np.array([[1.0, 3.0], [2.0, 4.0]]).T @ np.array([a / 2.0, -b])
        want = np.concatenate([g, mv, [max(min(-(a / 4), 0.98), -0.98)]])
        np.testing.assert_allclose(out, want, rtol=1e-15, atol=0)


@pytest.mark.gpu
def test_gpu_dynamics_against_exported_expression(golden_dir):
    """The HIP single-track step itself (RK4 x 1 over 0.08 s, what both the fused kernel and the SNMPC prologue integrate)
    against the reference's exported CasADi expression (snmpc_expr.npz), without the oracle in between: 264 random points
    placed on the stages of a small batch, Phi(x_k, u_k) read back as b_k + x_{k+1} through get_from_qp_in."""
    from tum_control_amd.solver import BatchedOcpSolver
    g = np.load(os.path.join(golden_dir, "snmpc_expr.npz"))
    pts = []
    for j in range(g["X"].shape[0]):
        X, F = g["X"][j].reshape(11, 8), g["F"][j].reshape(11, 8)
        if g["stop"][j] == 1.0:
            pts.append((X[0], g["U"][j], F[0]))                    # the nominal copy integrates on its own
        else:
            pts += [(X[i], g["U"][j], F[i]) for i in range(1, 11)]   # the sample copies do
    N = 40; B = (len(pts) + N - 1) // N
    Xi = np.zeros((B, N + 1, 8)); Ui = np.zeros((B, N, 2)); want = np.zeros((B, N, 8)); used = np.zeros((B, N), bool)
    for n, (x, u, f) in enumerate(pts):
        Xi[n // N, n % N] = x; Ui[n // N, n % N] = u; want[n // N, n % N] = f; used[n // N, n % N] = True
    Xi[~np.isfinite(Xi)] = 0.0
    for b in range(B):                                             # unused stages: a harmless state
        for k in range(N + 1):
            if k == N or not used[b, k]:
                Xi[b, k] = [0, 0, 0, 20, 0, 0, 0, 0]
    s = BatchedOcpSolver(N=N, dt=float(g["Ts"]), nsub=1, batch=B, store_qp_in=True)
    s.install_reference_ocp()
    s.set_x0(Xi[:, 0]); s.set_iterate(X=Xi, U=Ui)
    s.solve()
    worst = 0.0
    for k in range(N):
        phi = s.get_from_qp_in(k, "b").reshape(B, 8) + Xi[:, k + 1]
        for b in range(B):
            if used[b, k]:
                worst = max(worst, float(np.max(np.abs(phi[b] - want[b, k]) / (1.0 + np.abs(want[b, k])))))
    assert len(pts) == 264 and worst < 2e-5, worst


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nominal", "snmpc"])
def test_gpu_instrumented_kernels_agree(kind):
    """The four instantiations of the fused kernel (plain / instrumented x nominal / SNMPC) are the same arithmetic: a solve
    through the phase-timer entry point and one through the debug dump leave exactly the iterate of a plain solve. (The
    instrumented SNMPC instantiation is the largest one; a build of it once came out wrong while the others were right.)"""
    from tum_control_amd.solver import BatchedOcpSolver, CoupledSnmpcSolver
    from tum_control_amd.workloads import nominal_batch
    snm, stds, w, A = _pce()
    x0, yref = nominal_batch(8, N=40)
    res = {}
    from tum_control_amd import solver as _sv
    for mode in ("plain", "phases", "dump"):
        with _sv.dev_library():        # (the fused kernel lives in the development build; the pipeline has its own test below)
            if kind == "snmpc":
                s = CoupledSnmpcSolver(N=40, batch=8, Apce=A, uph=5, x0_offsets=snm.x0_offsets(w, stds), qp_warm_start=False)
            else:
                s = BatchedOcpSolver(N=40, batch=8, qp_warm_start=False)
        s.set_kernel("fused")
        s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
        if mode == "plain":
            s.solve()
        elif mode == "phases":
            s.profile_phases()
        else:
            s.debug_dump(0)
        res[mode] = (s.get_iterate()[1].copy(), s.get_stats("qp_iter").copy())
    for mode in ("phases", "dump"):
        np.testing.assert_array_equal(res[mode][1], res["plain"][1])
        np.testing.assert_array_equal(res[mode][0], res["plain"][0])


@pytest.mark.gpu
def test_gpu_pipeline_instrumented_kernel_agrees():
    """the same for the two instantiations of the pipeline's interior point kernel (plain / phase timers)"""
    from tum_control_amd.solver import BatchedOcpSolver
    from tum_control_amd.workloads import nominal_batch
    x0, yref = nominal_batch(24, N=40, seed=3)
    res = {}
    for mode in ("plain", "phases"):
        s = BatchedOcpSolver(N=40, batch=24)
        s.set_kernel("pipeline")
        s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
        if mode == "plain":
            s.solve()
        else:
            p = s.profile_phases()
            assert (p[:, [1, 2, 3, 5, 6, 7, 8, 9, 10, 11]] > 0).all()
            assert (p[:, 10] + p[:, 11] > p[:, 2]).all()          # the factorisation (slots 10, 11) dominates the residual phase
        res[mode] = (s.get_iterate()[1].copy(), s.get_stats("qp_iter").copy(), s.get_cost().copy())
    for i in range(3):
        np.testing.assert_array_equal(res["phases"][i], res["plain"][i])


@pytest.mark.gpu
@pytest.mark.parametrize("N,uph", [(38, 5), (40, 24), (38, 38), (48, 12), (12, 12), (38, 0)])
def test_gpu_snmpc_condensing_six_wavefronts_is_the_same_arithmetic(golden_dir, N, uph):
    """cond_wide_kernel<., true> (small batches) against cond_kernel<., true> on the nominal copy of the coupled SNMPC OCP:
    stages <= uph from the prologue's hand-over, the recursion behind them, |v| speed row -- every sum from the same operands
    in the same order, so three warm-started solves end bit-identical."""
    from tum_control_amd.solver import CoupledSnmpcSolver
    snm, stds, w, A = _pce()
    d = np.load(os.path.join(golden_dir, "kat0.npz"))
    poses = (0, 26, 30)
    B = len(poses)
    X0 = np.zeros((B, 11, 8)); Y = np.zeros((B, N + 1, 6))
    for j, i in enumerate(poses):
        X0[j] = snm.compute_x0dist(d["x0"][i].copy(), w, stds)
        yr = d["yref"][i]
        Y[j, :min(N, 38) + 1, :4] = yr[:min(N, 38) + 1]
        for k in range(39, N + 1):
            Y[j, k, :4] = 2 * Y[j, k - 1, :4] - Y[j, k - 2, :4]
    out = {}
    for name in ("cond-one-wavefront", "cond-six-wavefronts", "large-batch kernels"):
        s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=0.8)
        if name == "large-batch kernels":       # what a batch beyond the latency path runs: lin_kernel<true>, cond_kernel<., true>
            s.set_kernel("lin-lane-per-stage"); s.set_kernel("cond-one-wavefront")
        else:
            s.set_kernel(name)
        s.install_reference_ocp()
        s.constraints_set(0, "lbx", X0.reshape(B, -1)); s.constraints_set(0, "ubx", X0.reshape(B, -1))
        s.set_yref_all(Y); s.cold_start()
        for _ in range(3):
            assert s.solve() == 0
        Xn, U = s.get_iterate()
        XS = np.stack([np.atleast_2d(s.get(k, "x")) for k in (0, 1, max(uph, 1), N)])
        out[name] = (Xn, U, np.atleast_1d(s.get_cost()), s.get_stats("qp_iter"), XS)
    for p, q in zip(out["cond-one-wavefront"], out["cond-six-wavefronts"]):
        assert np.array_equal(p, q)
    # the eight-lane linearisation differs from the one-lane kernel by FMA contraction (3e-15 on A_k, B_k); three solves later:
    # (a real-time iteration that propagates the samples over many stages amplifies a perturbation of its linearisation by two
    #  orders of magnitude per solve, HISTORY.md (round-4 document, section 2): 4.7e-8 measured at N = uph = 12 after three solves)
    tol = 1e-8 if uph <= 5 else (1e-6 if uph <= 31 else 1e-5)
    for i in (0, 1, 4):
        assert np.abs(out["large-batch kernels"][i] - out["cond-one-wavefront"][i]).max() < tol


@pytest.mark.gpu
def test_gpu_snmpc_one_call_step_fans_the_state_out_on_the_device(golden_dir):
    """tum_ocp_step_async on a coupled SNMPC capsule: the 8 values of the estimated state go up with the step and the sample
    initial conditions are x0 + stds (.) w_s formed by the fan-out kernel -- the same numbers compute_x0dist forms on the host
    and the 88-value constraints_set path uploads; three warm control steps end bit-identical."""
    from tum_control_amd.solver import CoupledSnmpcSolver
    snm, stds, w, A = _pce()
    d = np.load(os.path.join(golden_dir, "kat0.npz"))
    N, uph, poses = 38, 5, (0, 26, 30)
    B = len(poses)
    offs = snm.x0_offsets(w, stds)
    Y = np.zeros((B, N + 1, 6)); x0 = np.stack([d["x0"][i] for i in poses])
    for j, i in enumerate(poses):
        Y[j, :, :4] = d["yref"][i][:N + 1]
    sols = []
    for _ in range(2):
        s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=0.8, x0_offsets=offs)
        s.install_reference_ocp()
        X0 = np.stack([snm.compute_x0dist(x0[j], w, stds) for j in range(B)])
        s.constraints_set(0, "lbx", X0.reshape(B, -1)); s.constraints_set(0, "ubx", X0.reshape(B, -1))
        s.set_yref_all(Y); s.cold_start()
        sols.append(s)
    a, b = sols
    rng = np.random.default_rng(3)
    for k in range(3):
        xk = x0 + k * 0.01 * rng.normal(size=x0.shape) * np.array([1, 1, .1, 1, .1, .02, .01, .1])
        Xk = np.stack([snm.compute_x0dist(xk[j], w, stds) for j in range(B)])
        a.constraints_set(0, "lbx", Xk.reshape(B, -1)); a.constraints_set(0, "ubx", Xk.reshape(B, -1))
        assert a.solve() == 0
        Xa, Ua = a.get_iterate()
        summ, Xb, Ub = b.step(x0=xk, yref=Y, with_iterate=True)
        assert (summ[:, 3] == 0).all()
        assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub)
        assert np.array_equal(np.atleast_2d(a.get(1, "x")), np.atleast_2d(b.get(1, "x")))
