import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
B, N = 4096, 40
out = {}
for seed in (0, 1, 2):
    x0, yref = nominal_batch(B, N=N, seed=seed)
    s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B)
    s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start(); s.solve()
    out[f"it{seed}"] = s.get_stats("qp_iter").copy()
    print(seed, out[f"it{seed}"].mean(), out[f"it{seed}"].max(), s.last_kernel_ms())
np.savez("gpurun_out/iters.npz", **out)
