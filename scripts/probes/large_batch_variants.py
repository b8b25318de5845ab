"""fused kernel against pipeline at large batches: iteration counts and iterates instance by instance (cold start)"""
import os, sys
os.environ.setdefault("TUM_NMPC_DEV", "1")      # (the fused kernel lives in the development build)
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
for B in (int(a) for a in (sys.argv[1:] or ["16384", "65536", "131072"])):
    x0, yref = nominal_batch(B, N=40, track_name="lvms", stride=7, seed=4321)
    res = {}
    for k in ("fused", "pipeline"):
        s = BatchedOcpSolver(N=40, batch=B); s.set_kernel(k); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start()
        st = s.solve(); it = s.get_stats("qp_iter").copy(); X, U = s.get_iterate()
        res[k] = (it, U.copy(), s.get_stats("status").copy())
        del s
    itf, Uf, sf = res["fused"]; itp, Up, sp = res["pipeline"]
    d = np.abs(Uf - Up).reshape(B, -1).max(axis=1)
    bad = np.where(itf != itp)[0]
    print(f"B {B}: mean qp_iter fused {itf.mean():.3f} pipeline {itp.mean():.3f}; instances with different iteration counts {bad.size}"
          f" (first {bad[:5]}, last {bad[-5:]}); max|dU| {d.max():.2e}, instances with |dU| > 1e-5: {(d > 1e-5).sum()}; status!=0 {int((sf != 0).sum())} / {int((sp != 0).sum())}")
    if bad.size:
        print("   index histogram of the differing instances (8 bins):", np.histogram(bad, bins=8, range=(0, B))[0])
