"""
The drop-in claim, tested: the reference's LITERAL AcadosOcpSolver call sequence -- what its controller classes issue per
control step, one synchronous call per stage and field -- against the logged acados outputs, and bit for bit against the one-call
step the mirrored classes use by default.

  nominal   Model_Predictive_Controller/Nominal_NMPC/NMPC_class.py:169-206 (solve), :243-246 (set_initial_state), :250-254 (reset),
            :290-317 (update_cost_function_weights): N x set(j,"yref",6) + set(N,"yref",4), solve(), get(0,"u"), N x get(j,"x"),
            get_cost(), get_stats('time_tot'|'sqp_iter'|'qp_iter'); constraints_set(0,"lbx"|"ubx",x0); cost_set(i,'W',...) per stage,
            cost_set(i,'zl'|'zu'|'Zl'|'Zu',...) per stage
  SNMPC     Model_Predictive_Controller/Stochastic_NMPC/SNMPC_class.py:181-214 (solve: additionally set(j,"p",...) per stage, the
            stacked state read with get(j,"x")[0:8]), :259-264 (set_initial_state: 8 (n_s+1) values)

A recording proxy around the solver object checks that the calls issued ARE that sequence (method, stage, field, length), so the
test cannot pass through a batch convenience.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIM = dict(Tp=3.04, Ts=0.02, Ts_MPC=0.08)


class Recorder:
    """forwards every AcadosOcpSolver method to the real solver and keeps (method, stage, field, n_values) of each call"""
    _METHODS = ("set", "get", "solve", "cost_set", "constraints_set", "get_cost", "get_stats", "reset")

    def __init__(self, solver):
        self._s, self.calls = solver, []

    def __getattr__(self, name):
        f = getattr(self._s, name)
        if name not in self._METHODS:
            return f

        def call(*a):
            n = int(np.size(a[2])) if len(a) > 2 else None
            self.calls.append((name,) + tuple(a[:2] if name not in ("get_stats",) else a[:1]) + ((n,) if n is not None else ()))
            return f(*a)
        return call

    def take(self):
        c, self.calls = self.calls, []
        return c


def _expected_solve_calls(N, with_p=None):
    e = []
    for j in range(N + 1):
        e.append(("set", j, "yref", 6 if j < N else 4))
        if with_p:
            e.append(("set", j, "p", with_p))
    e += [("solve",), ("get", 0, "u")] + [("get", j, "x") for j in range(N)]
    e += [("get_cost",), ("get_stats", "time_tot"), ("get_stats", "sqp_iter"), ("get_stats", "qp_iter")]
    return e


def _ref(y):
    return dict(pos_x=y[:, 0], pos_y=y[:, 1], ref_yaw=y[:, 2], ref_v=y[:, 3])


def test_literal_sequence_52_logged_cold_starts(golden_dir):
    """kat0.npz: the first solve of all 52 logged acados loops (26 weight sets x 2 tracks), each through the reference's own
    protocol on ONE solver object: update_cost_function_weights (per-stage cost_set), reset (reset + lbx_0/ubx_0 + N+1 x set 'x'),
    solve (the 2 N + 7 call sequence) -- 1e-6 relative against the logged u0 / x1 / cost."""
    from tum_control_amd.nmpc import Nonlinear_Model_Predictive_Controller as C
    d = dict(np.load(os.path.join(golden_dir, "kat0.npz")))
    c = C(sim_main_params=SIM, X0_MPC=d["x0"][0], call_pattern="acados")
    N = c.N
    rec = c.acados_solver = Recorder(c.acados_solver)
    for k in range(len(d["x0"])):
        c.update_cost_function_weights(d["params"][k])
        calls = rec.take()
        assert [x[:3] for x in calls[:N + 1]] == [("cost_set", i, "W") for i in range(N + 1)]
        assert len(calls) == (N + 1) + 4 * (N + 1)                      # W on every stage, four penalty vectors on every stage
        c.reset(d["x0"][k])
        calls = rec.take()
        assert calls == [("reset",), ("constraints_set", 0, "lbx", 8), ("constraints_set", 0, "ubx", 8)] + [("set", i, "x", 8) for i in range(N + 1)]
        u0, pred_X, stats = c.solve(_ref(d["yref"][k]))
        assert rec.take() == _expected_solve_calls(N)
        assert stats[4] == 0 and stats[2] == 1
        np.testing.assert_allclose(u0, d["u0"][k], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(pred_X[1], d["x1"][k], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(stats[0], d["cost"][k], rtol=1e-6)
        assert pred_X.shape == (N, 8) and stats[1] > 0


@pytest.mark.parametrize("name,tol", [("replay_lvms_0_0_450.npz", 2e-6), ("replay_monteblanco_0_0_400.npz", 2e-5)])
def test_literal_sequence_warm_loop_equals_one_call_step_and_the_log(golden_dir, name, tol):
    """A logged warm-started loop (set_initial_state + solve every control step) through the literal sequence AND through the
    one-call step, two controllers side by side: every step bit-identical between the two, and both within `tol` of acados' log."""
    from tum_control_amd.nmpc import Nonlinear_Model_Predictive_Controller as C
    d = dict(np.load(os.path.join(golden_dir, name)))
    a = C(sim_main_params=SIM, X0_MPC=d["x0"][0], call_pattern="acados")
    b = C(sim_main_params=SIM, X0_MPC=d["x0"][0], call_pattern="step")
    rec = a.acados_solver = Recorder(a.acados_solver)
    a.update_cost_function_weights(d["params"]); b.update_cost_function_weights(d["params"])
    rec.take()
    worst = 0.0
    for i in range(len(d["x0"])):
        if i > 0:
            a.set_initial_state(d["x0"][i]); b.set_initial_state(d["x0"][i])
            assert rec.take() == [("constraints_set", 0, "lbx", 8), ("constraints_set", 0, "ubx", 8)]
        ref = _ref(d["yref"][i])
        ua, Xa, sa = a.solve(ref)
        ub, Xb, sb = b.solve(ref)
        assert rec.take() == _expected_solve_calls(a.N)
        assert np.array_equal(ua, ub) and np.array_equal(Xa, Xb)
        assert sa[0] == sb[0] and sa[3] == sb[3] and sa[4] == sb[4] == 0
        worst = max(worst, np.abs(ua - d["u0"][i]).max(), np.abs(Xa[1] - d["x1"][i]).max())
        assert abs(sa[0] - d["cost"][i]) <= 1e-4 * max(1.0, abs(d["cost"][i]))
    assert worst < tol


def test_literal_sequence_snmpc_vs_oracle_and_one_call_step(golden_dir):
    """The stochastic controller's sequence (SNMPC_class.py:181-214: yref AND the 102-value parameter vector on every stage, the
    stacked 88-value state read back per stage) on the coupled OCP: against the CPU oracle, and bit for bit against the one-call step,
    over a cold start and warm control steps with moving initial states."""
    from oracle import oracle as orc
    from tum_control_amd import config, snmpc as snm
    from tum_control_amd.snmpc import Stochastic_Nonlinear_Model_Predictive_Controller as C
    d = dict(np.load(os.path.join(golden_dir, "kat0.npz")))
    x0, yref = d["x0"][0], d["yref"][0]
    a = C(sim_main_params=SIM, X0_MPC=x0, call_pattern="acados")
    b = C(sim_main_params=SIM, X0_MPC=x0, call_pattern="step")
    N = a.N
    rec = a.acados_solver = Recorder(a.acados_solver)
    m = config.MPC
    o = orc.OracleSnmpcOcp(N=N, dt=a.Tp / N, Apce=a.A, uph=5)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    o.yref[:, :4] = yref; o.cold_start(snm.compute_x0dist(x0, a.w_samples, a.stds))
    ref = _ref(yref)
    np_ = a.A.size + 2
    nxs = 8 * (a.n_samples + 1)
    x = x0.copy()
    for k in range(6):
        if k:
            a.set_initial_state(x); b.set_initial_state(x)
            assert rec.take() == [("constraints_set", 0, "lbx", nxs), ("constraints_set", 0, "ubx", nxs)]
            o.set_initial_state(snm.compute_x0dist(x, a.w_samples, a.stds))
        ua, Xa, sa = a.solve(ref)
        ub, Xb, sb = b.solve(ref)
        assert rec.take() == _expected_solve_calls(N, with_p=np_)
        assert o.solve() == 0 and sa[4] == 0 and sb[4] == 0
        assert np.array_equal(ua, ub) and np.array_equal(Xa, Xb) and sa[0] == sb[0] and sa[3] == sb[3]
        np.testing.assert_allclose(ua, o.U[0], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(Xa, o.X[:N, 0], rtol=1e-7, atol=2e-8)
        np.testing.assert_allclose(sa[0], o.cost, rtol=1e-7)
        x = Xa[1] + 1e-3 * np.array([1.0, -1.0, 0.1, 0.5, 0.05, 0.01, 0.0, 0.0]) * (k + 1)


def test_step_delivers_its_own_results_after_an_abandoned_request(golden_dir):
    """tum_ocp_results_wait hands out the OLDEST outstanding request. A request someone left behind on the capsule (a results_async
    never waited for) must not make every later step() return the previous solve's results: step() drains first."""
    from tum_control_amd.nmpc import Nonlinear_Model_Predictive_Controller as C
    d = dict(np.load(os.path.join(golden_dir, "replay_lvms_0_0_450.npz")))
    a = C(sim_main_params=SIM, X0_MPC=d["x0"][0]); b = C(sim_main_params=SIM, X0_MPC=d["x0"][0])
    a.update_cost_function_weights(d["params"]); b.update_cost_function_weights(d["params"])
    for i in range(6):
        a.set_initial_state(d["x0"][i]); b.set_initial_state(d["x0"][i])
        if i == 2:
            b._solver.results_async(True)                 # abandoned: nobody waits for it
            assert b._solver.results_outstanding() == 1
        ua, Xa, sa = a.solve(_ref(d["yref"][i])); ub, Xb, sb = b.solve(_ref(d["yref"][i]))
        assert b._solver.results_outstanding() == 0
        assert np.array_equal(ua, ub) and np.array_equal(Xa, Xb) and sa[0] == sb[0]


def test_pending_initial_state_is_flushed_for_outside_users(golden_dir):
    """set_initial_state of the one-call pattern parks x0 for the next step; code that then drives `controller.acados_solver`
    directly (the reference's other pattern) must see it: reading the attribute flushes."""
    from tum_control_amd.nmpc import Nonlinear_Model_Predictive_Controller as C
    d = dict(np.load(os.path.join(golden_dir, "replay_lvms_0_0_450.npz")))
    a = C(sim_main_params=SIM, X0_MPC=d["x0"][0]); b = C(sim_main_params=SIM, X0_MPC=d["x0"][0], call_pattern="acados")
    y = np.zeros((a.N + 1, 6)); y[:, :4] = d["yref"][0]
    for c in (a, b):
        c.update_cost_function_weights(d["params"])
        c.set_initial_state(d["x0"][5])
        s = c.acados_solver
        s.set_yref_all(y)
        assert s.solve() == 0
    Xa, Ua = a.acados_solver.get_iterate(); Xb, Ub = b.acados_solver.get_iterate()
    assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub)
    np.testing.assert_allclose(Xa[0, 0], d["x0"][5], rtol=1e-14, atol=1e-15)          # (x_0 + (x0 - x_0): the initial-value embedding, 1 ulp)


def _pair(golden_dir, B=3, N=38):
    from tum_control_amd.solver import BatchedOcpSolver
    from tum_control_amd.workloads import nominal_batch
    x0, yref = nominal_batch(B, N=N, seed=77)
    sol = []
    for _ in range(2):
        s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B)
        s.install_reference_ocp()
        s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
        sol.append(s)
    return sol[0], sol[1], x0, yref


def test_getters_after_a_synchronous_solve_read_the_pinned_slabs_and_say_the_same(golden_dir):
    """Small capsule: solve() leaves summary / X / U in pinned slabs and get / get_cost / get_stats are host copies. Held against a
    second capsule that solves asynchronously (no slabs: every getter is a device read), on a cold start and two warm steps."""
    a, b, x0, yref = _pair(golden_dir)
    N, B = a.N, a.batch
    for k in range(3):
        if k:
            x = Xa[:, 1] + 0.01 * k
            for s in (a, b):
                s.constraints_set(0, "lbx", x); s.constraints_set(0, "ubx", x)
        st = a.solve()
        b.solve_async(); b.synchronize()
        assert st == int(b.get_stats("status").max()) == 0
        Xa, Ua = a.get_iterate(); Xb, Ub = b.get_iterate()
        assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub)
        for j in (0, 1, N // 2, N):
            assert np.array_equal(a.get(j, "x"), b.get(j, "x")) and np.array_equal(a.get(j, "x"), Xa[:, j])
        assert np.array_equal(a.get(0, "u"), b.get(0, "u")) and np.array_equal(a.get(N - 1, "u"), Ub[:, N - 1])
        assert np.array_equal(a.get_cost(), b.get_cost())
        assert np.array_equal(a.get_stats("qp_iter"), b.get_stats("qp_iter")) and np.array_equal(a.get_stats("status"), b.get_stats("status"))
        ta, tb = a.get_stats("time_tot"), b.get_stats("time_tot")          # device clock against HIP events
        assert 2e-5 < ta < 5e-3 and 2e-5 < tb < 5e-3 and 0.3 < ta / tb < 3.0


def test_slabs_are_dropped_when_the_iterate_changes(golden_dir):
    a, b, x0, yref = _pair(golden_dir)
    N, B = a.N, a.batch
    assert a.solve() == 0
    X, U = a.get_iterate()
    v = np.arange(8.0) + 0.5
    a.set(3, "x", v)                                             # a setter on the iterate: the next getter must read the device
    assert np.array_equal(a.get(3, "x"), np.tile(v, (B, 1))) and np.array_equal(a.get(4, "x"), X[:, 4])
    assert a.solve() == 0
    a.set(2, "u", np.array([0.25, -0.125]))
    assert np.array_equal(a.get(2, "u"), np.tile([0.25, -0.125], (B, 1)))
    assert a.solve() == 0
    a.cold_start()
    assert np.array_equal(a.get(N, "x"), x0) and not a.get(0, "u").any()
    assert a.solve() == 0
    a.reset()
    assert not a.get(5, "x").any()
    # an asynchronous solve behind a synchronous one: the slabs of the first must not answer for the second
    a.set_x0(x0); a.cold_start(); assert a.solve() == 0
    X1, U1 = a.get_iterate()
    a.solve_async(); a.synchronize()
    X2, U2 = a.get_iterate()
    b.solve(); b.solve()
    Xb, Ub = b.get_iterate()
    assert np.array_equal(X2, Xb) and np.array_equal(U2, Ub) and not np.array_equal(X1, X2)


def test_per_stage_setters_wait_in_the_shadow_in_order(golden_dir):
    """set(j, "yref") / constraints_set(0, "lbx") of a small capsule are parked in pinned host memory and uploaded with the next solve.
    Whatever else writes the same device arrays in between must keep its place in the order: a whole-horizon set, a device upload,
    a one-call step; stages nobody set keep what the device holds."""
    import torch
    a, b, x0, yref = _pair(golden_dir)
    N, B = a.N, a.batch
    rng = np.random.default_rng(5)
    y2 = yref + rng.normal(size=yref.shape) * np.array([0.3, 0.3, 0.01, 0.2, 0, 0])
    y3 = yref + rng.normal(size=yref.shape) * np.array([0.3, 0.3, 0.01, 0.2, 0, 0])
    # (1) only the odd stages per stage, the even ones keep the device values
    for j in range(1, N + 1, 2):
        a.set(j, "yref", y2[:, j, :6 if j < N else 4])
    mix = yref.copy(); mix[:, 1::2] = y2[:, 1::2]
    mix[:, N, 4:] = 0
    b.set_yref_all(mix)
    assert a.solve() == b.solve() == 0
    assert all(np.array_equal(p, q) for p, q in zip(a.get_iterate(), b.get_iterate()))
    # (2) per-stage setters, THEN a whole-horizon set: the later call wins on every stage
    for j in range(N + 1):
        a.set(j, "yref", y2[:, j, :6 if j < N else 4])
    a.set_yref_all(y3); b.set_yref_all(y3)
    assert a.solve() == b.solve() == 0
    assert all(np.array_equal(p, q) for p, q in zip(a.get_iterate(), b.get_iterate()))
    # (3) a device upload, THEN per-stage setters of three stages
    t = torch.as_tensor(y2.reshape(B, -1), device="cuda:0").contiguous()
    a.put_device("yref", t.data_ptr()); torch.cuda.synchronize()
    for j in (0, 7, N):
        a.set(j, "yref", y3[:, j, :6 if j < N else 4])
    mix = y2.copy(); mix[:, [0, 7]] = y3[:, [0, 7]]; mix[:, N, :4] = y3[:, N, :4]
    b.set_yref_all(mix)
    assert a.solve() == b.solve() == 0
    assert all(np.array_equal(p, q) for p, q in zip(a.get_iterate(), b.get_iterate()))
    # (4) x0 in the shadow, then a one-call step that brings its own x0: the step's wins; then an asynchronous solve uses a set x0
    xa = x0 + 0.01
    a.constraints_set(0, "lbx", x0 + 5.0)
    sa, Xs, Us = a.step(x0=xa, yref=y3)
    b.set_x0(xa); b.set_yref_all(y3); b.solve()
    assert np.array_equal(Xs, b.get_iterate()[0])
    a.constraints_set(0, "lbx", x0 + 0.02); a.solve_async(); a.constraints_set(0, "lbx", x0 + 0.03); a.synchronize()
    b.set_x0(x0 + 0.02); b.solve()
    assert np.array_equal(a.get_iterate()[0], b.get_iterate()[0])
    a.solve(); b.set_x0(x0 + 0.03); b.solve()
    assert np.array_equal(a.get_iterate()[0], b.get_iterate()[0])


def test_snmpc_eight_value_initial_state_through_the_shadow(golden_dir):
    """A coupled SNMPC capsule with registered sample offsets takes the 8 nominal values at lbx_0 / ubx_0 and fans them out on the
    device; on a small capsule that setter is parked in the pinned shadow like the nominal one. Against a capsule that is given the
    stacked 88 values (uploaded at once), over warm steps; then the stacked form again on the first capsule (fan-out switched off)."""
    from tum_control_amd import config, snmpc as snm
    from tum_control_amd.solver import CoupledSnmpcSolver
    stds = np.asarray(config.MPC["stds"], dtype=float)
    w = snm.hammersley_normal(10, 3); A = snm.pce_matrix(w, snm.alpha_generation(3, 2)); offs = snm.x0_offsets(w, stds)
    d = dict(np.load(os.path.join(golden_dir, "kat0.npz")))
    N, B = 38, 2
    x0 = np.stack([d["x0"][0], d["x0"][30]]); Y = np.zeros((B, N + 1, 6)); Y[0, :, :4] = d["yref"][0]; Y[1, :, :4] = d["yref"][30]
    stack = lambda x: np.stack([snm.compute_x0dist(x[j], w, stds) for j in range(B)]).reshape(B, -1)
    sol = []
    for _ in range(2):
        s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=5, gamma=0.8, x0_offsets=offs)
        s.install_reference_ocp()
        s.constraints_set(0, "lbx", stack(x0)); s.constraints_set(0, "ubx", stack(x0)); s.set_yref_all(Y); s.cold_start()
        sol.append(s)
    a, b = sol
    for k in range(4):
        xk = x0 + 0.01 * k * np.array([1, -1, 0.1, 1, 0.1, 0.02, 0.01, 0.1])
        if k < 3:
            a.constraints_set(0, "lbx", xk); a.constraints_set(0, "ubx", xk)                    # 8 values: shadow + fan-out kernel
        else:
            a.constraints_set(0, "lbx", stack(xk)); a.constraints_set(0, "ubx", stack(xk))      # back to explicit samples
        b.constraints_set(0, "lbx", stack(xk)); b.constraints_set(0, "ubx", stack(xk))
        assert a.solve() == b.solve() == 0
        assert all(np.array_equal(p, q) for p, q in zip(a.get_iterate(), b.get_iterate()))
        for j in (0, 3, 9, N):
            assert np.array_equal(a.get(j, "x"), b.get(j, "x"))          # the stacked 88-value state (cached read-back on both)
