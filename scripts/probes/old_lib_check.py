#!/usr/bin/env python3
"""Do the four instantiations of the fused kernel of an OLDER build of libtumnmpc.so agree bit for bit?
   python scripts/probes/old_lib_check.py /path/to/libtumnmpc.so
(used while chasing the wrong instrumented build of round 1: HISTORY.md, "An unexplained build failure"). Symbols the old
library lacks are replaced by no-ops so that today's binding loads it."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd import solver, snmpc as snm, config


class _Dummy:
    argtypes = None; restype = None
    def __call__(self, *a):
        return 0


class _Tolerant:
    def __init__(self, lib):
        object.__setattr__(self, "_l", lib)
    def __getattr__(self, n):
        try:
            return getattr(self._l, n)
        except AttributeError:
            d = _Dummy(); object.__setattr__(self, n, d); return d


_real = ctypes.CDLL
solver.ctypes.CDLL = lambda p, *a, **k: _Tolerant(_real(p, *a, **k))
solver.LIB_PATH = sys.argv[1]
from tum_control_amd.workloads import nominal_batch
stds = np.asarray(config.MPC["stds"]); w = snm.hammersley_normal(10, 3)
A = snm.pce_matrix(w, snm.alpha_generation(3, 2))
x0, yref = nominal_batch(8, N=40)
ok = True
for kind in ("nominal", "snmpc"):
    res = {}
    for mode in ("plain", "phases", "dump"):
        if kind == "snmpc":
            s = solver.CoupledSnmpcSolver(N=40, batch=8, Apce=A, uph=5, x0_offsets=snm.x0_offsets(w, stds))
        else:
            s = solver.BatchedOcpSolver(N=40, batch=8)
        s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
        if mode == "plain":
            s.solve()
        elif mode == "phases":
            s.profile_phases()
        else:
            s.debug_dump(0)
        res[mode] = (s.get_iterate()[1].copy(), s.get_stats("qp_iter").copy())
    for mode in ("phases", "dump"):
        same = np.array_equal(res[mode][0], res["plain"][0]) and np.array_equal(res[mode][1], res["plain"][1])
        ok &= same
        print(f"{kind:8s} {mode:7s} vs plain: {'identical' if same else 'DIFFERENT'}; qp_iter {res[mode][1].tolist()} (plain {res['plain'][1].tolist()}); max|dU| {np.abs(res[mode][0] - res['plain'][0]).max():.3e}")
print("ALL IDENTICAL" if ok else "MISMATCH")
