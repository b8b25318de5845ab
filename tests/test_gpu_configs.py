"""
BASELINE.json configs 3, 4 and 5 at their per-GPU sizes through the C-ABI (-m gpu), plus the entry points added for them
(device-side PCE moments, bounds snapshot / restore, device-to-device inputs), the per-stage parameter vector of the SNMPC
OCP at the C level, and the recovery from a failed solve in the closed loops.

Protocol per config (the oracle cannot solve 16384 instances in seconds): the oracle on a strided subset of the instances,
size-independent properties (bitwise determinism, independence of the batch composition, group reductions against numpy)
at the full size.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 40


def _mk(B, **kw):
    from tum_control_amd.solver import BatchedOcpSolver
    s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B, **kw)
    s.install_reference_ocp()
    return s


def _oracle():
    from oracle.oracle import OracleOcp
    from tum_control_amd import config
    m = config.MPC
    o = OracleOcp(N, 0.08, 3)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    return o


def _check_subset_against_oracle(s, x0, yref, idx, tol=1e-6):
    X, U = s.get_iterate()
    it = s.get_stats("qp_iter")
    u0, X1, st = _oracle().solve_batch_cold(x0[idx], yref[idx], 8)
    assert (st[:, 2] == 0).all()
    same = it[idx] == st[:, 1]          # (an instance whose termination test sits on the tolerance edge may stop one iteration apart)
    assert same.mean() > 0.98 and np.abs(it[idx] - st[:, 1]).max() <= 1
    eu = np.abs(U[idx, 0] - u0).max(axis=1); ex = np.abs(X[idx, 1] - X1).max(axis=1)
    assert eu[same].max() < tol and ex[same].max() < tol
    assert eu.max() < 5e-5 and ex.max() < 5e-5
    np.testing.assert_allclose(s.get_cost()[idx][same], st[same, 0], rtol=1e-7)
    if (~same).any():      # the instances that stopped one iteration apart, against the oracle with the GPU's count imposed
        d = idx[~same]
        u0f, X1f, stf = _oracle().solve_batch_cold(x0[d], yref[d], 8, force_iter=it[d])
        assert np.abs(U[d, 0] - u0f).max() < tol and np.abs(X[d, 1] - X1f).max() < tol
        np.testing.assert_allclose(s.get_cost()[d], stf[:, 0], rtol=1e-7)


def test_config3_sigma_points_full_size():
    """configs[2]: 16384 scenarios = 1024 poses x (nominal + 15 Hammersley sigma points), Monteblanco; cold-start SQP-RTI
    and the PCE moments of x_1 per scenario group (K6), host and device flavour."""
    import torch
    from tum_control_amd.snmpc import alpha_generation, hammersley_normal, pce_matrix, x0_offsets
    from tum_control_amd.workloads import config_groups, sigma_point_groups
    from tum_control_amd import config
    P, S1 = 1024, 16
    x0, yref, g = config_groups(3, 0, P, P, N=N)
    assert g == S1 and len(x0) == P * S1
    s = _mk(P * S1)
    # the fan-out kernel (compute_x0dist as a batch axis) produces the same x0 as the host generator
    w = hammersley_normal(15, 3)
    off = x0_offsets(w, config.MPC["stds"])
    pose, yg, _, _ = sigma_point_groups(0, P, P, off, N=N)
    s.set_x0_fanout(pose, off)
    s.set_yref_all(yref); s.cold_start()
    np.testing.assert_array_equal(s.get(0, "x"), x0)
    assert s.solve() == 0
    assert (s.get_stats("status") == 0).all()
    X, U = s.get_iterate()
    _check_subset_against_oracle(s, x0, yref, np.arange(5, P * S1, 257))
    # PCE moments at full size against numpy, host-pointer and device-pointer entry points
    A = pce_matrix(w, alpha_generation(3, 2))
    c = np.einsum("ls,psm->plm", A, X[:, 1].reshape(P, S1, 8)[:, 1:])
    mean, var = s.pce_moments("x", 1, A)
    np.testing.assert_allclose(mean, c[:, 0], atol=1e-10)
    np.testing.assert_allclose(var, (c[:, 1:] ** 2).sum(axis=1), atol=1e-10)
    s.pce_attach(A)
    mv = torch.zeros((2, P, 8), dtype=torch.float64, device="cuda")
    s.pce_moments_device("x", 1, mv[0].data_ptr(), mv[1].data_ptr())
    s.synchronize()
    assert np.array_equal(mv[0].cpu().numpy(), mean) and np.array_equal(mv[1].cpu().numpy(), var)
    # the propagated uncertainty is what the sigma points say: the spread of vlong after one stage stays near its std
    assert 0.5 < np.sqrt(var[:, 3]).mean() / config.MPC["stds"][3] < 1.2
    # determinism and independence of the batch composition: the second half of the groups alone gives the same bits
    s.cold_start(); s.solve()
    X2, U2 = s.get_iterate()
    assert np.array_equal(X, X2) and np.array_equal(U, U2)
    h = _mk(P * S1 // 2)
    h.set_x0(x0[P * S1 // 2:]); h.set_yref_all(yref[P * S1 // 2:]); h.cold_start(); assert h.solve() == 0
    Xh, Uh = h.get_iterate()
    assert np.array_equal(Xh, X[P * S1 // 2:]) and np.array_equal(Uh, U[P * S1 // 2:])


def test_config4_monte_carlo_full_size():
    """configs[3]: the per-GPU share (16384 scenarios = 1024 poses x 16) of the 131072-scenario Monte-Carlo job on LVMS:
    rank 5 of 8 by the group-aligned sharding; oracle on a strided subset, determinism, permutation of whole groups."""
    from tum_control_amd.sharding import shard_groups
    from tum_control_amd.workloads import CONFIGS, config_groups
    world, rank = 8, 5
    gt = world * CONFIGS[4]["groups_per_gpu"]
    g_lo, g_hi, b_lo, b_hi = shard_groups(gt, 16, world, rank)
    assert (g_hi - g_lo, b_hi - b_lo) == (1024, 16384) and b_lo == 16 * g_lo
    x0, yref, _ = config_groups(4, g_lo, g_hi, gt, N=N)
    B = len(x0)
    s = _mk(B)
    s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
    assert s.solve() == 0
    st = s.get_stats("status")
    assert (st == 0).all()
    X, U = s.get_iterate()
    _check_subset_against_oracle(s, x0, yref, np.arange(3, B, 251))
    # every scenario of a group starts from the pose's position / heading and only differs in (vlong, vlat, yawrate)
    d = x0.reshape(1024, 16, 8) - x0.reshape(1024, 16, 8)[:, :1]
    assert (d[:, :, [0, 1, 2, 6, 7]] == 0).all() and (np.abs(d[:, 1:, 3]) > 0).all()
    # permuting whole groups permutes the results (bitwise); so does the schedule
    perm = np.random.default_rng(3).permutation(1024)
    idx = (perm[:, None] * 16 + np.arange(16)[None]).reshape(-1)
    s.set_x0(x0[idx]); s.set_yref_all(yref[idx]); s.cold_start(); assert s.solve() == 0
    Xp, Up = s.get_iterate()
    assert np.array_equal(Xp, X[idx]) and np.array_equal(Up, U[idx])
    s.set_schedule(False); s.cold_start(); s.solve(); s.set_schedule(True)
    Xn, Un = s.get_iterate()
    assert np.array_equal(Xn, Xp) and np.array_equal(Un, Up)


def test_config5_r2_full_size():
    """configs[4]: 4096 R2NMPC instances (the per-GPU share of 32768) on Modena: solve, covariance back-off (K7), solve with
    the tightened bounds. The one-shot path (tum_ocp_r2_backoff between two solves) against numpy / the oracle on a subset;
    the attached path with bounds snapshot / restore (what bench.py --config 5 times) bitwise against the one-shot path."""
    from oracle.oracle import h_con
    from tum_control_amd.r2nmpc import r2_setup
    from tum_control_amd.workloads import config_groups
    from tum_control_amd import config
    B = 4096
    x0, yref, _ = config_groups(5, 0, B, 8 * B, N=N)
    m, veh = config.MPC, config.VEH
    S0, BWB = r2_setup(m["stds"], 0.08)
    uph, dmin, dmax = int(m["uncertainty_propagation_horizon"]), veh["delta_f_min"], veh["delta_f_max"]
    a = _mk(B, store_qp_in=True)
    a.set_x0(x0); a.set_yref_all(yref); a.cold_start()
    assert a.solve() == 0
    X1, U1 = a.get_iterate()
    idx = np.arange(7, B, 273)
    _check_subset_against_oracle(a, x0, yref, idx)
    bo = a.r2_backoff(S0, BWB, uph, dmin, dmax, 1.0, return_backoffs=True)
    A = np.stack([a.get_from_qp_in(k, "A") for k in range(uph)], axis=1)
    for b in idx:
        Sig = S0.copy(); bd = bh = 0.0
        for k in range(uph):
            if k > 0:
                _, g = h_con(X1[b, k])
                bd = np.sqrt(Sig[6, 6]); bh = np.sqrt(g @ Sig @ g)
                assert abs(bo[b, k, 0] - bd) < 1e-12 and abs(bo[b, k, 1] - bh) < 1e-10
            Sig = A[b, k] @ Sig @ A[b, k].T + BWB
        assert np.abs(bo[b, uph:, 0] - bd).max() < 1e-12 and np.abs(bo[b, uph:, 1] - bh).max() < 1e-10
    assert (bo[:, 1:, 1] > 0).all() and bo[:, 1:, 1].max() < 0.5
    assert a.solve() == 0
    X2, U2 = a.get_iterate()
    for b in idx[:6]:
        o = _oracle(); o.cold_start(x0[b]); o.yref[:] = yref[b]; assert o.solve() == 0
        o.lbx[1:N] = dmin + bo[b, 1:, 0]; o.ubx[1:N] = dmax - bo[b, 1:, 0]; o.uh[1:N] = 1.0 - bo[b, 1:, 1]
        assert o.solve() == 0
        assert np.abs(U2[b] - o.U).max() < 1e-6 and np.abs(X2[b] - o.X).max() < 1e-6
    # the attached path: two steps of (restore nominal bounds, cold start, solve + K7, solve + K7)
    c = _mk(B, store_qp_in=True)
    c.set_x0(x0); c.set_yref_all(yref)
    c.r2_attach(S0, BWB, uph, dmin, dmax, 1.0)
    c.bounds_snapshot()
    for _ in range(2):
        c.bounds_restore(); c.cold_start()
        assert c.constraints_get(3, "uh").min() == 1.0
        c.solve_async(); c.solve_async(); c.synchronize()
        Xc, Uc = c.get_iterate()
        assert np.array_equal(Xc, X2) and np.array_equal(Uc, U2)
        assert (c.get_stats("status") == 0).all()
    np.testing.assert_allclose(c.constraints_get(3, "uh"), 1.0 - a.r2_backoff(S0, BWB, uph, dmin, dmax, 1.0, return_backoffs=True)[:, 3, 1], atol=1e-14)


def test_put_device_inputs():
    """per-instance inputs straight from caller-owned HBM (the fresh-batch leg of bench.py)"""
    import torch
    from tum_control_amd.workloads import nominal_batch
    B = 64
    x0, yref = nominal_batch(B, N=N, seed=9)
    a = _mk(B); a.set_x0(x0); a.set_yref_all(yref); a.cold_start(); assert a.solve() == 0
    Xa, Ua = a.get_iterate()
    b = _mk(B)
    tx, ty = torch.from_numpy(x0).cuda(), torch.from_numpy(yref.copy()).cuda()
    torch.cuda.synchronize()
    b.put_device("x0", tx.data_ptr()); b.put_device("yref", ty.data_ptr()); b.cold_start(); assert b.solve() == 0
    Xb, Ub = b.get_iterate()
    assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub)
    with pytest.raises(Exception, match="unknown field"):
        b.put_device("nope", tx.data_ptr())


def test_gpu_per_stage_parameters():
    """set(stage, "p", [A_pce.flatten(), risk_parameter, stop_flag]) at the C level (SNMPC_class.py:124,185,193): the stop
    flags define the uncertainty propagation horizon, A_pce and the risk parameter are shared by the stages."""
    from tum_control_amd import config
    from tum_control_amd import snmpc as snm
    from tum_control_amd.solver import CoupledSnmpcSolver
    from tum_control_amd.workloads import nominal_batch
    stds = np.asarray(config.MPC["stds"]); w = snm.hammersley_normal(10, 3)
    A = snm.pce_matrix(w, snm.alpha_generation(3, 2)); off = snm.x0_offsets(w, stds)
    Nn, B = 38, 5
    x0, yref = nominal_batch(B, N=Nn, seed=4)

    def run(s):
        s.install_reference_ocp()
        s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
        assert s.solve() == 0
        return s.get_iterate()

    def setp(s, A_, gam, uph):
        for k in range(Nn + 1):
            s.set(k, "p", np.concatenate([A_.flatten(), [gam], [1.0 if k >= uph else 0.0]]))

    ref7 = run(CoupledSnmpcSolver(N=Nn, dt=0.08, batch=B, Apce=A, uph=7, gamma=0.8, x0_offsets=off))
    ref3g = run(CoupledSnmpcSolver(N=Nn, dt=0.08, batch=B, Apce=0.9 * A, uph=3, gamma=0.7, x0_offsets=off))
    s = CoupledSnmpcSolver(N=Nn, dt=0.08, batch=B, Apce=A, uph=5, gamma=0.8, x0_offsets=off)
    base = run(s)
    assert not np.array_equal(base[1], ref7[1])
    setp(s, A, 0.8, 7)                       # longer horizon than the capsule was attached with: buffers grow
    got = run(s)
    assert np.array_equal(got[0], ref7[0]) and np.array_equal(got[1], ref7[1])
    setp(s, 0.9 * A, 0.7, 3)                 # other PCE matrix, risk level and horizon
    got = run(s)
    assert np.array_equal(got[0], ref3g[0]) and np.array_equal(got[1], ref3g[1])
    # patterns the stacked model is not built for are refused at the next solve
    s.set(10, "p", np.concatenate([0.9 * A.flatten(), [0.7], [0.0]]))
    with pytest.raises(Exception, match="stop_flag pattern"):
        s.solve()
    setp(s, 0.9 * A, 0.7, 3)
    s.set(2, "p", np.concatenate([0.9 * A.flatten(), [0.6], [0.0]]))
    with pytest.raises(Exception, match="risk parameter"):
        s.solve()
    setp(s, 0.9 * A, 0.7, 3)
    assert s.solve() == 0
    with pytest.raises(Exception, match="mismatching dimension"):
        s.set(1, "p", np.zeros(5))
    nom = _mk(2)
    with pytest.raises(Exception, match="not an SNMPC capsule"):
        nom._chk(nom._L.tum_ocp_set(nom._h, 0, b"p", np.zeros(3).ctypes.data_as(nom._L.tum_ocp_set.argtypes[3]), 3, 0, 2, 0), "set")


@pytest.mark.parametrize("controller", ["nominal", "r2", "snmpc"])
def test_closed_loop_recovers_from_a_failed_solve(controller):
    """main.py:59-61: a failed solve is followed by MPC.reintialize_solver(x_next) -- a fresh solver cold-started at the
    state the failed solve started from. Here one vehicle's iterate is poisoned (NaN) in the middle of a closed loop: that
    solve returns status 4, the loop applies the stale control, re-initialises the instance and carries on; the all-device
    loop (recovery inside plant_advance_kernel) and the host loop (ClosedLoopBatch._reinitialise) agree step for step."""
    from tum_control_amd.closed_loop import ClosedLoopBatch
    B, n1, n2, Nn = 3, 6, 8, 38
    loops = []
    for on_device in (False, True):
        cl = ClosedLoopBatch("monteblanco", batch=B, N=Nn, Tp=3.04, controller=controller, on_device=on_device,
                             log_capacity=n1 + n2, idx_start=100)
        cl.run(n1)
        X, U = cl.solver.get_iterate()
        X[1, 3:7, :] = np.nan
        cl.solver.set_iterate(X, U)
        if controller == "r2":
            assert cl.solver.constraints_get(3, "uh").max() < 1.0       # tightened by the solves so far
        if on_device and controller == "r2":
            cl.dev.run(1)
            uh = cl.solver.constraints_get(3, "uh")
            assert uh[1] == 1.0 and uh[0] < 1.0 and uh[2] < 1.0         # nominal bounds again for the failed instance only
            lg = cl.run(n2 - 1)
        else:
            lg = cl.run(n2)
        loops.append(lg)
    host, dev = loops
    dbg = dev["simSolverDebug"]
    assert dbg.shape[0] == n1 + n2
    assert dbg[n1, 1, 4] == 4 and (dbg[n1, [0, 2], 4] == 0).all()
    assert (dbg[:n1, :, 4] == 0).all() and (dbg[n1 + 1:, :, 4] == 0).all()      # back to normal from the next step on
    for k in ("simU", "CiLX", "MPC_SimX"):
        assert np.isfinite(dev[k]).all()
        np.testing.assert_allclose(dev[k], host[k], rtol=1e-7, atol=1e-8, err_msg=k)
    np.testing.assert_array_equal(dev["simSolverDebug"][:, :, 4], host["simSolverDebug"][:, :, 4])
    # the stale control was applied at the failed step: u0 of that step equals the previous step's
    np.testing.assert_array_equal(dev["simU"][n1, 1], dev["simU"][n1 - 1, 1])
    # and the other vehicles never noticed
    ref = ClosedLoopBatch("monteblanco", batch=1, N=Nn, Tp=3.04, controller=controller, on_device=True,
                          log_capacity=n1 + n2, idx_start=100).run(n1 + n2)
    np.testing.assert_allclose(dev["CiLX"][:, 0], ref["CiLX"][:, 0], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("iterate", [False, True])
def test_bench_under_torchrun_world_size_1_rccl(iterate):
    """(iterate: --gather-iterate, the whole iterate X, U -- 3264 B per instance, 53 MB for this shard -- rides in the same rooted
    gather) The multi-GPU code path of bench.py on hardware, at world size 1: launched by torch.distributed.run like the driver's
    N > 1 runs (one rank per GPU), so `init_process_group("nccl")` (= RCCL), the device-packed result slab, the rooted
    `dist.gather` inside the timed region and the max-over-ranks `all_reduce` all run on the GPU -- everything of SURVEY 8(e)
    except a second rank. Config 4 (Monte-Carlo scenarios, LVMS) in its strong-scaling form with the 16384-instance share of
    one GPU as the global batch."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--config", "4", "--scaling", "strong",
           "--global-batch", "16384", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-schedule-legs"] + (["--gather-iterate"] if iterate else [])
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 16384 and out["config"]["batch_per_gpu"] == 16384
    assert isinstance(out["gather_ms_per_step"], float) and 0.0 < out["gather_ms_per_step"] < 50.0      # the RCCL gather ran and was timed
    assert out["status_ok_frac"] == 1.0
    assert out["value"] > 1e5
    assert out["config"]["collective_backend"] == "nccl"          # (nccl IS RCCL on ROCm)
    assert out["config"]["gather_iterate"] == iterate and out["config"]["gather_bytes_per_rank"] == 8 * 16384 * (5 + (41 * 8 + 40 * 2 if iterate else 0))


def test_bench_self_launched_world_size_1_rccl():
    """`python bench.py --gpus N` with no launcher around it starts its own ranks (bench.self_launch); BENCH_SELF_LAUNCH=1 forces that
    branch at N = 1, so the whole self-launched multi-GPU path runs on the MI355X -- child process per GPU, MASTER_ADDR 127.0.0.1 and a
    free port, init_process_group("nccl") (= RCCL), the rooted gather and the per-rank all_gather inside / behind the timed region --
    minus a second rank. The line says how it was launched and what the collective library reports."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SELF_LAUNCH="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_SELF_LAUNCHED"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
           "--no-schedule-legs", "--no-host-legs", "--no-other-configs"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["launch"].startswith("self-launched") and out["n_gpus"] == 1
    assert out["config"]["collective_backend"] == "nccl" and out["config"]["collective_world_size"] == 1
    assert out["config"]["batch_per_gpu"] == 4096 and out["status_ok_frac"] == 1.0
    assert isinstance(out["gather_ms_per_step"], float) and 0.0 < out["gather_ms_per_step"] < 50.0
    assert len(out["per_rank"]["value"]) == 1 and abs(out["per_rank"]["value"][0] / out["value"] - 1.0) < 0.05
    assert out["value"] > 1e6


def test_bench_line_carries_the_other_configurations():
    """the default N = 1 line of bench.py carries a bounded leg for each of the other BASELINE configurations and for the coupled SNMPC OCP
    (`other_configs`: what the driver times next to the headline): all five present, every instance solved, rates in the range the
    separate runs of the same configurations give (profiles/r06_configs.jsonl, r06_snmpc_bench.txt)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_SELF_LAUNCH", "BENCH_SELF_LAUNCHED"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-schedule-legs", "--no-host-legs"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][0])
    oc = out["other_configs"]
    assert sorted(oc) == ["3", "4", "5", "snmpc_uph38", "snmpc_uph5"]
    for cid, batch in (("3", 16384), ("4", 16384), ("5", 4096)):
        leg = oc[cid]
        assert "error" not in leg, leg
        assert leg["batch"] == batch and leg["status_ok_frac"] == 1.0 and leg["value"] > 2e6 and 0.2 < leg["roofline"]["frac"] < 0.6, leg
    for key, lo in (("snmpc_uph5", 2.5e6), ("snmpc_uph38", 1.2e6)):
        leg = oc[key]
        assert "error" not in leg, leg
        assert leg["single_capsule"]["status_ok_frac"] == 1.0 and leg["three_capsules"]["status_ok_frac"] == 1.0
        assert leg["single_capsule"]["value"] > lo and leg["value"] >= 0.95 * leg["single_capsule"]["value"], leg
    assert out["launch"] == "single process" and len(out["per_rank"]["value"]) == 1


def test_results_on_the_host_through_pinned_slabs():
    """tum_ocp_results_async / _wait: the summary (u0, cost, status, qp_iter) and the whole iterate of a batch arrive in the
    capsule's pinned host slabs behind an event, equal to what the synchronous getters return; four capsules in a ring (its cap) with
    four different batches in flight hand every batch's OWN results back (streaming.SolverRing.request_results / take_results)."""
    from tum_control_amd.streaming import SolverRing
    from tum_control_amd.workloads import nominal_batch
    B = 1536
    batches = [nominal_batch(B, N=N, seed=100 + k) for k in range(4)]
    ring = SolverRing(4, lambda i: _mk(B))
    assert ring.take_results(0) is None
    for k in range(4):
        slot, s = ring.acquire()
        s.set_x0(batches[k][0]); s.set_yref_all(batches[k][1]); s.cold_start(); s.solve_async()
        ring.request_results(slot, with_iterate=True)
    got = dict(ring.drain())
    assert sorted(got) == [0, 1, 2, 3]
    for k in range(4):
        summ, X, U = got[k]
        s = ring[k]
        Xr, Ur = s.get_iterate()
        assert np.array_equal(X, Xr) and np.array_equal(U, Ur)
        assert np.array_equal(summ[:, :2], Ur[:, 0]) and np.array_equal(summ[:, 2], s.get_cost())
        assert np.array_equal(summ[:, 3], s.get_stats("status")) and np.array_equal(summ[:, 4], s.get_stats("qp_iter"))
        assert (summ[:, 3] == 0).all()
    assert not np.array_equal(got[0][0], got[1][0])
    # summary alone: no iterate slabs are handed out
    slot, s = ring.acquire()
    s.cold_start(); s.solve_async(); ring.request_results(slot)
    summ, X, U = ring.take_results(slot)
    assert X is None and U is None and np.array_equal(summ, got[slot][0])
    # two requests may be outstanding on a capsule (two sets of slabs, used in turn), taken oldest first; a third is refused
    s2 = _mk(4)
    with pytest.raises(Exception, match="results_wait"):
        s2.results_wait()
    x0, yr = nominal_batch(4, N=N, seed=7)
    s2.set_x0(x0); s2.set_yref_all(yr); s2.cold_start(); s2.solve_async(); s2.results_async(True)
    s2.solve_async(); s2.results_async(True)                        # a second real-time iteration behind the first
    with pytest.raises(Exception, match="two requests outstanding"):
        s2.results_async()
    (a, Xa, Ua), (b, Xb, Ub) = s2.results_wait(), s2.results_wait()
    Xr, Ur = s2.get_iterate()
    assert np.array_equal(Xb, Xr) and np.array_equal(Ub, Ur) and not np.array_equal(Ua, Ub)
    assert a.ctypes.data != b.ctypes.data and np.array_equal(b[:, :2], Ur[:, 0])


@pytest.mark.gpu
def test_one_call_control_step_equals_setters_solve_getters():
    """tum_ocp_step_async (x0 and yref through pinned staging, the solve and the results request behind them) + results_wait
    against the acados-style sequence constraints_set / set_yref_all / solve / getters: a warm-started sequence of control
    steps with a new x0 and a new reference every step ends bit-identical; null inputs keep what the capsule has; two steps
    may be outstanding and a third is refused."""
    from tum_control_amd.workloads import nominal_batch
    B, K = 5, 4
    seq = [nominal_batch(B, N=N, seed=300 + k) for k in range(K)]
    a, b = _mk(B), _mk(B)
    for s in (a, b):
        s.set_x0(seq[0][0]); s.set_yref_all(seq[0][1]); s.cold_start()
    for k in range(K):
        x0, yr = seq[k]
        a.set_x0(x0); a.set_yref_all(yr)
        assert a.solve() == 0
        Xa, Ua = a.get_iterate()
        summ, Xb, Ub = b.step(x0=x0, yref=yr, with_iterate=True)
        assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub)
        assert np.array_equal(summ[:, :2], Ua[:, 0]) and np.array_equal(summ[:, 2], a.get_cost())
        assert np.array_equal(summ[:, 3], a.get_stats("status")) and np.array_equal(summ[:, 4], a.get_stats("qp_iter"))
    # the device time of a step (read from the device's wall clock by its first and last kernel) against the events of solve()
    t_step, t_solve = b.get_stats("time_tot"), a.get_stats("time_tot")
    assert 2e-5 < t_step < 5e-3 and 2e-5 < t_solve < 5e-3 and 0.3 < t_step / t_solve < 3.0
    with pytest.raises(Exception, match="time_ipm"):
        b.get_stats("time_ipm")                     # (a step leaves the events around the interior point kernel out)
    # null inputs: another real-time iteration on the same data
    assert a.solve() == 0
    summ, Xb, Ub = b.step(with_iterate=True)
    Xa, Ua = a.get_iterate()
    assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub)
    # two steps in flight (each with its own staging area and slabs), a third refused
    b.step_async(x0=seq[1][0], yref=seq[1][1]); b.step_async(x0=seq[2][0], yref=seq[2][1])
    with pytest.raises(Exception, match="two requests outstanding"):
        b.step_async()
    r1, r2 = b.results_wait(), b.results_wait()
    a.set_x0(seq[1][0]); a.set_yref_all(seq[1][1]); a.solve(); U1 = a.get_iterate()[1].copy()
    a.set_x0(seq[2][0]); a.set_yref_all(seq[2][1]); a.solve(); U2 = a.get_iterate()[1]
    assert np.array_equal(r1[2], U1) and np.array_equal(r2[2], U2)
    with pytest.raises(Exception, match="expected"):
        b.step(x0=np.zeros(3))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["derivatives", "state_estimation", "both"])
def test_disturbed_closed_loop_device_against_host(which, tmp_path):
    """sim_step's disturbance simulation (Utils/SimulationMode_main_class.py:121-143) in the device loop -- the realisation played back
    by plant_advance_kernel: a second plant step with xdot + w for what the estimator sees, the estimation error on top, the TRUE state
    still the undisturbed step -- against the host loop that applies the same realisation with the numpy plant; then the log file."""
    from tum_control_amd.closed_loop import ClosedLoopBatch, DisturbanceModel, LOG_KEYS
    m = DisturbanceModel(simulate_disturbances=which != "state_estimation", simulate_state_estimation=which != "derivatives")
    n, B = 60, 3
    w, e = m.draw(n - 10, B, seed=5)          # (shorter than the run: the last ten steps are undisturbed)
    logs = {}
    for dev in (False, True):
        cl = ClosedLoopBatch("lvms", batch=B, N=38, Tp=3.04, on_device=dev, log_capacity=n, disturbances=(w, e))
        logs[dev] = cl.run(n)
    for f in ("simU", "CiLX", "MPC_SimX", "simREF"):
        np.testing.assert_allclose(logs[True][f][:25], logs[False][f][:25], rtol=1e-8, atol=1e-8, err_msg=f)
        np.testing.assert_allclose(logs[True][f], logs[False][f], rtol=1e-5, atol=1e-5, err_msg=f)
    assert (logs[True]["simSolverDebug"][:, :, 4] == 0).all()
    # the disturbance reaches the controller (the loops differ from the undisturbed one) but not the plant's own integration:
    clean = ClosedLoopBatch("lvms", batch=B, N=38, Tp=3.04, on_device=True, log_capacity=n).run(n)
    assert np.abs(clean["simU"][1:] - logs[True]["simU"][1:]).max() > 1e-3
    np.testing.assert_array_equal(clean["CiLX"][:2], logs[True]["CiLX"][:2])          # step 0 is solved before any disturbance is seen
    # the vehicles of a batch see different realisations
    assert np.abs(logs[True]["simU"][5:, 0] - logs[True]["simU"][5:, 1]).max() > 1e-4
    paths = cl.save(str(tmp_path / "{}.npz"))
    assert len(paths) == B
    f = np.load(paths[1])
    assert set(f.files) == set(LOG_KEYS)
    assert f["simU"].shape == (n - 1, 2) and f["CiLX"].shape == (n, 7) and f["sim_disturbance_derivatives"].shape == (n, 7) and f["t"].shape == (n - 1,)
    if w is not None:
        np.testing.assert_array_equal(f["sim_disturbance_derivatives"][:n - 10], w[:, 1])
        assert not f["sim_disturbance_derivatives"][n - 10:].any()
    else:
        assert not f["sim_disturbance_derivatives"].any()
    np.testing.assert_array_equal(f["simU"], logs[True]["simU"][:n - 1, 1])
    assert (f["CiLX"][:, 2] >= 0).all() and (f["CiLX"][:, 2] < 2 * np.pi).all()


@pytest.mark.gpu
def test_bound_device_inputs_are_read_in_place():
    """tum_ocp_bind_device: the capsule uses the caller's device arrays as its x0 / yref arrays -- no copy. Same results as the copying upload
    (tum_ocp_put_device), rebinding to another resident batch switches batches, setters write THROUGH to the caller's memory while bound, and
    unbinding hands the capsule's own arrays back with their old contents."""
    import torch
    from tum_control_amd.workloads import nominal_batch
    B = 300
    b0, b1 = nominal_batch(B, N=N, seed=41), nominal_batch(B, N=N, seed=42)
    dev = [[torch.as_tensor(np.ascontiguousarray(v), device="cuda:0") for v in b] for b in (b0, b1)]
    a, c = _mk(B), _mk(B)
    for s in (a, c):
        s.set_x0(b0[0]); s.set_yref_all(b0[1]); s.cold_start(); assert s.solve() == 0          # the capsules' own arrays hold batch 0
    for k in (1, 0, 1):
        x, y = dev[k]
        a.put_device("x0", x.data_ptr()); a.put_device("yref", y.data_ptr())
        c.bind_device("x0", x.data_ptr()); c.bind_device("yref", y.data_ptr())
        for s in (a, c):
            s.cold_start(); assert s.solve() == 0
        assert all(np.array_equal(p, q) for p, q in zip(a.get_iterate(), c.get_iterate()))
    # setters write through while bound
    x1 = b1[0] + 0.01
    c.set_x0(x1); c.synchronize()
    assert np.array_equal(dev[1][0].cpu().numpy(), x1)
    a.set_x0(x1)
    for s in (a, c):
        s.cold_start(); assert s.solve() == 0
    assert all(np.array_equal(p, q) for p, q in zip(a.get_iterate(), c.get_iterate()))
    # unbound: the capsule's own arrays again, still holding batch 0
    c.bind_device("x0", None); c.bind_device("yref", None)
    a.set_x0(b0[0]); a.set_yref_all(b0[1])
    for s in (a, c):
        s.cold_start(); assert s.solve() == 0
    assert all(np.array_equal(p, q) for p, q in zip(a.get_iterate(), c.get_iterate()))
    with pytest.raises(Exception, match="field must be"):
        c.bind_device("X", dev[0][0].data_ptr())
