#!/usr/bin/env python3
"""Interleaved A/B of builds of the library in ONE process on BASELINE configs[1] (4096 x N=40, cold start, repeated batch with the
exact longest-first schedule): usage ab2.py libA.so libB.so ... ("shipped" = tum-control_amd/libtumnmpc.so). Prints, per build,
min / median of the pipeline time and of ipm_kernel over ROUNDS interleaved rounds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd import solver as sv
from tum_control_amd.workloads import nominal_batch

N, B, ROUNDS = int(os.environ.get("N", "40")), int(os.environ.get("B", "4096")), int(os.environ.get("ROUNDS", "30"))
x0, yref = nominal_batch(B, N=N)
names = sys.argv[1:]
sol = []
for n in names:
    p = sv.LIB_PATH if n == "shipped" else os.path.abspath(n)
    sv.load_library(p)
    old = sv._default_path; sv._default_path = p
    try:
        s = sv.BatchedOcpSolver(N=N, batch=B, store_qp_in=False)
    finally:
        sv._default_path = old
    try:
        s.install_reference_ocp()
    except Exception:          # a library from before per-stage W (round 4): cost_set at stage 0 sets all stages there
        sv.ALL_STAGES = 0
        try:
            s.install_reference_ocp()
        finally:
            sv.ALL_STAGES = -1
    s.set_x0(x0); s.set_yref_all(yref)
    for _ in range(3):
        s.cold_start(); s.solve()
    sol.append(s)
ms = [[] for _ in sol]; ipm = [[] for _ in sol]
for r in range(ROUNDS):
    for i, s in enumerate(sol):
        s.cold_start(); s.solve(); ms[i].append(s.last_kernel_ms()); ipm[i].append(1e3 * s.get_stats("time_ipm"))
for n, s, m, t in zip(names, sol, ms, ipm):
    rest = np.array(m) - np.array(t)
    print(f"{n:40s} qp_iter {s.get_stats('qp_iter').mean():.3f} pipeline ms min {min(m):.4f} med {np.median(m):.4f} | ipm ms min {min(t):.4f} med {np.median(t):.4f} | lin + cond + expand ms min {rest.min():.4f} med {np.median(rest):.4f}", flush=True)
