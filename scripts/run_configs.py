#!/usr/bin/env python3
"""Runs the per-GPU share of every BASELINE.json config on one MI355X and prints one line per config
(throughput of the hot path on that workload + sanity checks). bench.py stays the contract for configs[1];
this script documents that the other configs run through the same C-ABI.

  configs[0]  nominal NMPC closed loop, Monteblanco, single instance (plumbing)      -> closed_loop.ClosedLoopBatch(batch=1)
  configs[1]  nominal batch 4096, perturbed x0, N=40                                   -> bench.py workload
  configs[2]  SNMPC sigma points, 16384 scenarios = 1024 poses x (1 + 15)             -> snmpc.ScenarioSNMPC (+ PCE moments, K6)
  configs[3]  Monte-Carlo 131072 scenarios over 8 GPUs = 16384 per GPU, LVMS         -> scenario fan-out with random offsets
  configs[4]  R2NMPC 32768 over 8 GPUs = 4096 per GPU, Modena, two solves + back-off  -> r2nmpc.ReducedRobustifiedNMPC (K7)
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd import config
from tum_control_amd.closed_loop import ClosedLoopBatch
from tum_control_amd.snmpc import ScenarioSNMPC
from tum_control_amd.r2nmpc import ReducedRobustifiedNMPC
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch, scenario_batch


def timed(solver, reps=5, prepare=None):
    ms = []
    for _ in range(reps):
        if prepare:
            prepare()
        solver.solve()
        ms.append(solver.last_kernel_ms())
    return float(np.median(ms))


def main():
    N = 40
    # configs[0]
    t0 = time.perf_counter()
    cl = ClosedLoopBatch("monteblanco", batch=1, N=38, Tp=3.04)
    log = cl.run(250)
    wall = time.perf_counter() - t0
    dbg = log["simSolverDebug"][:, 0]
    print(f"configs[0] closed loop 250 steps (5 s of driving), batch 1: all status 0 = {bool((dbg[:, 4] == 0).all())}, "
          f"mean qp_iter {dbg[:, 3].mean():.1f}, mean kernel {1e3 * dbg[:, 1].mean():.3f} ms/solve, wall {wall:.1f} s incl. host planner/plant")
    # configs[0] as BASELINE states it (5000 steps), planner / plant / estimator as device kernels; then the same loop
    # for the 26 weight sets of the BO sweep and for 4096 vehicles at once
    for B, steps in ((1, 5000), (26, 5000), (4096, 500)):
        cl = ClosedLoopBatch("monteblanco", batch=B, N=38, Tp=3.04, on_device=True, log_capacity=steps)
        t0 = time.perf_counter()
        lg = cl.run(steps)
        wall = time.perf_counter() - t0
        dbg = lg["simSolverDebug"]
        print(f"configs[0] on-device closed loop, batch {B}, {steps} steps ({steps * 0.02:.0f} s of driving): wall {wall:.2f} s = "
              f"{1e3 * wall / steps:.3f} ms/step, {B * steps / wall:,.0f} closed-loop solves/s, status 0 {(dbg[:, :, 4] == 0).mean():.4f}, "
              f"mean qp_iter {dbg[:, :, 3].mean():.2f}, max |lateral speed| {np.abs(lg['CiLX'][:, :, 4]).max():.2f} m/s")
        del cl
    # configs[1]
    x0, yref = nominal_batch(4096, N=N)
    s = BatchedOcpSolver(N=N, batch=4096); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref)
    ms = timed(s, prepare=s.cold_start)
    print(f"configs[1] nominal batch 4096: {ms:.3f} ms -> {4096 / ms * 1e3:,.0f} solves/s, status0 {(s.get_stats('status') == 0).mean():.4f}, qp_iter {s.get_stats('qp_iter').mean():.2f}")
    del s
    # the same at the reference's own horizon (N = 38, Tp = 3.04 s) for comparability with its logs (0.65-1.2 ms per acados call)
    x0, yref = nominal_batch(4096, N=38)
    s = BatchedOcpSolver(N=38, batch=4096); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref)
    ms = timed(s, prepare=s.cold_start)
    print(f"configs[1] at N=38: {ms:.3f} ms -> {4096 / ms * 1e3:,.0f} solves/s, qp_iter {s.get_stats('qp_iter').mean():.2f}")
    # warm-started RTI steps on the same batch (iterate kept, x0 moved to the predicted next state)
    X, U = s.get_iterate(); s.set_x0(X[:, 1]); ws = []
    for _ in range(5):
        s.solve(); ws.append(s.last_kernel_ms()); X, U = s.get_iterate(); s.set_x0(X[:, 1])
    print(f"configs[1] at N=38, warm RTI steps: {np.median(ws):.3f} ms -> {4096 / np.median(ws) * 1e3:,.0f} solves/s, qp_iter {s.get_stats('qp_iter').mean():.2f}")
    del s
    # configs[2]
    P = 1024
    sn = ScenarioSNMPC(P, n_samples=15, N=N)
    x0s, yrefs, S1 = scenario_batch(P, sn.offsets, N=N)
    st, u0n, mean, var = sn.solve(x0s[::S1], yrefs[::S1])
    ms = timed(sn.solver, prepare=sn.solver.cold_start)
    print(f"configs[2] SNMPC sigma points {P} poses x {S1} = {P * S1}: {ms:.3f} ms -> {P * S1 / ms * 1e3:,.0f} solves/s, status {st}, "
          f"PCE std of x1[vlong] mean {np.sqrt(var[:, 3]).mean():.4f}")
    del sn
    # configs[3]: Monte-Carlo scenarios, LVMS, 16384 per GPU (1024 poses x 16 draws), seed 4321 + rank
    rng = np.random.default_rng(4321)
    stds = np.asarray(config.MPC["stds"])
    off = rng.standard_normal((15, 8)) * stds
    x0m, yrefm, S1 = scenario_batch(1024, off, N=N, track_name="lvms", pose_stride=7)
    s = BatchedOcpSolver(N=N, batch=len(x0m)); s.install_reference_ocp(); s.set_x0(x0m); s.set_yref_all(yrefm)
    ms = timed(s, prepare=s.cold_start)
    print(f"configs[3] Monte-Carlo LVMS {len(x0m)} per GPU: {ms:.3f} ms -> {len(x0m) / ms * 1e3:,.0f} solves/s, status0 {(s.get_stats('status') == 0).mean():.4f}")
    del s
    # configs[4]: R2NMPC, Modena, 4096 per GPU, two consecutive solves with the back-off kernel in between
    x0r, yrefr = nominal_batch(4096, N=N, track_name="modena", seed=777)
    r2 = ReducedRobustifiedNMPC(batch=4096, N=N)
    r2.solver.set_x0(x0r); r2.solver.set_yref_all(yrefr); r2.solver.cold_start()
    t0 = time.perf_counter(); st1 = r2.solve(); k1 = r2.solver.last_kernel_ms(); st2 = r2.solve(); k2 = r2.solver.last_kernel_ms(); wall = time.perf_counter() - t0
    print(f"configs[4] R2NMPC Modena 4096 per GPU: solve1 {k1:.3f} ms, solve2 (tightened) {k2:.3f} ms, status {st1},{st2}, "
          f"uh at stage 3: min {r2.solver.constraints_get(3, 'uh').min():.4f}, wall for both incl. K7 {1e3 * wall:.1f} ms")


if __name__ == "__main__":
    main()
