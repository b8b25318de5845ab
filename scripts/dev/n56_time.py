import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch  # noqa
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
B = 4096
for N in (40, 48, 50, 56):
    x0, yref = nominal_batch(B, N=N, seed=40 + N)
    s = BatchedOcpSolver(N=N, batch=B); s.install_reference_ocp(); s.set_kernel("time-ipm"); s.set_x0(x0); s.set_yref_all(yref)
    ms, ipm = [], []
    for r in range(8):
        s.cold_start(); s.solve(); ms.append(s.last_kernel_ms()); ipm.append(1e3 * s.get_stats("time_ipm"))
    print(f"N {N}: 4096 cold starts (repeated batch): solve {np.median(ms[2:]):.3f} ms = {B / np.median(ms[2:]) / 1e3:.2f} M solves/s, ipm_kernel {np.median(ipm[2:]):.3f} ms, qp_iter {s.get_stats('qp_iter').mean():.2f}, status0 {(s.get_stats('status') == 0).mean():.4f}")
    del s
# the mirrored controller class at Tp = 4.0 s (N = 50)
from tum_control_amd import nmpc, config
cfg = config.default_config()
sim = dict(cfg["sim"]); sim["Tp"] = 4.0
try:
    import inspect
    print("controller signature:", inspect.signature(nmpc.Nonlinear_Model_Predictive_Controller.__init__))
except Exception as e:
    print(e)
