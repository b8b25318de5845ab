#!/usr/bin/env python3
"""Throughput of the coupled SNMPC OCP on one MI355X: batches of the reference's stochastic controller
solve (prologue + fused kernel + epilogue, timed together with HIP events), cold start and warm real-time iterations,
next to the dense CPU restatement (oracle, one thread) on a few instances and the reference's logged acados time
(about 6.1 ms per SNMPC solve)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd import config, snmpc as snm
from tum_control_amd.solver import CoupledSnmpcSolver
from tum_control_amd.workloads import nominal_batch


def main():
    B = int(os.environ.get("SN_BATCH", 4096))
    stds = np.asarray(config.MPC["stds"], dtype=float)
    w = snm.hammersley_normal(10, 3)
    A = snm.pce_matrix(w, snm.alpha_generation(3, 2))
    offs = snm.x0_offsets(w, stds)
    cases = ((38, 5), (40, 5), (38, 9), (40, 15), (40, 24), (38, 38))               # (38, 38): UPH = Tp as in the ACC24 campaign of the reference
    if os.environ.get("SN_ONLY"):                                                   # e.g. SN_ONLY=38,38
        cases = (tuple(int(v) for v in os.environ["SN_ONLY"].split(",")),)
    for N, uph in cases:
        x0, yref = nominal_batch(B, N=N)
        X0 = np.concatenate([x0[:, None, :], x0[:, None, :] + offs[None]], axis=1)          # (B, 11, 8)
        s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=config.MPC["gamma"])
        s.install_reference_ocp()
        if os.environ.get("SN_PROLOGUE"):          # prologue-cols | prologue-passes (default: the library's choice)
            s.set_kernel(os.environ["SN_PROLOGUE"])
        s.constraints_set(0, "lbx", X0.reshape(B, -1)); s.constraints_set(0, "ubx", X0.reshape(B, -1))
        s.set_yref_all(yref)
        cold = []
        for _ in range(5):
            s.cold_start(); st = s.solve(); cold.append(s.last_kernel_ms())
        it_c = s.get_stats("qp_iter").mean(); ok_c = (s.get_stats("status") == 0).mean()
        warm = []
        for _ in range(5):
            st = s.solve(); warm.append(s.last_kernel_ms())
        it_w = s.get_stats("qp_iter").mean()
        mc, mw = float(np.median(cold)), float(np.median(warm))
        print(f"coupled SNMPC N={N} uph={uph} ns=10 batch {B}: cold {mc:.3f} ms -> {B / mc * 1e3:,.0f} solves/s (qp_iter {it_c:.2f}, status0 {ok_c:.4f}); "
              f"warm RTI {mw:.3f} ms -> {B / mw * 1e3:,.0f} solves/s (qp_iter {it_w:.2f})")
        if N == 38:
            from oracle import oracle as orc
            m = config.MPC
            t = []
            for j in range(4):
                o = orc.OracleSnmpcOcp(N=N, dt=0.08, Apce=A, uph=uph)
                o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
                o.yref[:] = yref[j]; o.cold_start(X0[j])
                t0 = time.perf_counter(); o.solve(); t.append(time.perf_counter() - t0)
            print(f"  CPU restatement (88-state condensing that skips the zero blocks, one thread): {1e3 * np.median(t):.1f} ms per solve; "
                  f"reference acados SNMPC: about 6.1 ms per solve (its logs)")
        del s
        if (N, uph) in ((38, 5), (38, 38)) and int(os.environ.get("SN_STREAMS", 3)) > 1:
            # several batches in flight, as the headline of bench.py: S capsules on their own streams, a cold start + solve
            # enqueued on each in turn (independent batches: scenario sweeps have nothing to wait for), wall clock over K rounds
            from tum_control_amd.streaming import SolverRing
            S = int(os.environ.get("SN_STREAMS", 3))

            def mk(_):
                c = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=config.MPC["gamma"])
                c.install_reference_ocp()
                c.constraints_set(0, "lbx", X0.reshape(B, -1)); c.constraints_set(0, "ubx", X0.reshape(B, -1))
                c.set_yref_all(yref)
                return c
            ring = SolverRing(S, mk)
            for _ in range(2 * S):
                _, c = ring.acquire(); c.cold_start(); c.solve_async()
            ring.synchronize()
            K = 8 * S
            t0 = time.perf_counter()
            for _ in range(K):
                _, c = ring.acquire(); c.cold_start(); c.solve_async()
            ring.synchronize()
            dt_ = time.perf_counter() - t0
            ok = min((c.get_stats("status") == 0).mean() for c in ring)
            print(f"  {S} capsules in flight (cold start + solve each, {K} batches): {1e3 * dt_ / K:.3f} ms per batch -> {B * K / dt_:,.0f} solves/s (status0 {ok:.4f})")
            del ring
    if os.environ.get("SN_ONLY"):
        return
    # the SNMPC controller in closed loop, planner / plant / estimator / x0 fan-out as device kernels
    from tum_control_amd.closed_loop import ClosedLoopBatch
    for Bc, steps in ((1, 1000), (4096, 200)):
        cl = ClosedLoopBatch("monteblanco", batch=Bc, N=38, Tp=3.04, controller="snmpc", on_device=True, log_capacity=steps)
        t0 = time.perf_counter(); lg = cl.run(steps); wall = time.perf_counter() - t0
        dbg = lg["simSolverDebug"]
        print(f"SNMPC closed loop on the device, batch {Bc}, {steps} steps: {1e3 * wall / steps:.3f} ms/step, {Bc * steps / wall:,.0f} closed-loop solves/s, "
              f"status 0 {(dbg[:, :, 4] == 0).mean():.4f}, mean qp_iter {dbg[:, :, 3].mean():.2f}")
        del cl


if __name__ == "__main__":
    main()
