#!/usr/bin/env python3
"""30 cold solves of BASELINE configs[1] (4096 x N = 40) with ONE build of the library (argument: path or "shipped"), for rocprofv3 --kernel-trace --stats:
per-kernel averages of an experimental build (exp_libs/, scripts/dev/build_exp_lib.sh) next to the shipped one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from tum_control_amd import solver as sv
from tum_control_amd.workloads import nominal_batch
N, B = 40, int(os.environ.get("B", "4096"))
n = sys.argv[1]
p = sv.LIB_PATH if n == "shipped" else os.path.abspath(n)
sv.load_library(p); sv._default_path = p
s = sv.BatchedOcpSolver(N=N, batch=B, store_qp_in=False)
s.install_reference_ocp()
x0, yref = nominal_batch(B, N=N)
s.set_x0(x0); s.set_yref_all(yref)
for _ in range(int(os.environ.get("REPS", "30"))):
    s.cold_start(); s.solve()
print(n, "mean qp_iter", float(s.get_stats("qp_iter").mean()))
