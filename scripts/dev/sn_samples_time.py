#!/usr/bin/env python3
"""Coupled SNMPC OCP at other sample counts / expansion degrees than the shipped 10 / 2 (round 6: up to 32 samples and 32 PCE terms; beyond 16 of
either the 32-wide instantiations of the column-slot prologue and the epilogue run). 4096 instances, N = 38, cold start + solve, kernel time."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd import config, snmpc as snm
from tum_control_amd.solver import CoupledSnmpcSolver
from tum_control_amd.workloads import nominal_batch

B, N = 4096, 38
stds = np.asarray(config.MPC["stds"], dtype=float)
x0, yref = nominal_batch(B, N=N)
for ns, deg, uph in ((10, 2, 5), (16, 2, 5), (20, 2, 5), (24, 3, 5), (32, 3, 5), (24, 3, 18)):
    w = snm.hammersley_normal(ns, 3)
    A = snm.pce_matrix(w, snm.alpha_generation(3, deg))
    offs = snm.x0_offsets(w, stds)
    X0 = np.concatenate([x0[:, None, :], x0[:, None, :] + offs[None]], axis=1)
    s = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=config.MPC["gamma"])
    s.install_reference_ocp()
    s.constraints_set(0, "lbx", X0.reshape(B, -1)); s.constraints_set(0, "ubx", X0.reshape(B, -1))
    s.set_yref_all(yref)
    t = []
    for _ in range(5):
        s.cold_start(); s.solve(); t.append(s.last_kernel_ms())
    ok = (s.get_stats("status") == 0).mean()
    mc = float(np.median(t))
    print(f"samples {ns:2d} degree {deg} ({A.shape[0]:2d} PCE terms) uph {uph:2d} batch {B}: {mc:.3f} ms -> {B / mc * 1e3:,.0f} solves/s (status0 {ok:.4f}, qp_iter {s.get_stats('qp_iter').mean():.2f})")
    del s
