import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tum_control_amd.solver import BatchedOcpSolver
from tum_control_amd.workloads import nominal_batch
N, B = 40, 512
x0, yref = nominal_batch(B, N=N, seed=9)
def mk(w):
    s = BatchedOcpSolver(N=N, batch=B, qp_warm_start=w); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref); s.cold_start(); assert s.solve() == 0; return s
a, b = mk(True), mk(False)
for k in range(6):
    Xa, Ua = a.get_iterate()
    b.set_iterate(Xa, Ua)
    a.set_x0(Xa[:, 1]); b.set_x0(Xa[:, 1])
    sa, sb = a.solve(), b.solve()
    dU = np.abs(a.get_iterate()[1] - b.get_iterate()[1]).max(axis=(1, 2))
    ia, ib = a.get_stats("qp_iter"), b.get_stats("qp_iter")
    qa, qb = a.get_stats("qp_status"), b.get_stats("qp_status")
    j = int(dU.argmax())
    print(f"step {k}: status {sa} {sb}; mean it warm {ia.mean():.2f} cold {ib.mean():.2f}; max it {ia.max()} {ib.max()}; capped warm {(qa==1).sum()} cold {(qb==1).sum()}; "
          f"median dU {np.median(dU):.1e} max {dU.max():.1e} at inst {j}: it {ia[j]} vs {ib[j]}, qp_status {qa[j]} vs {qb[j]}, res warm {a.get_stats('res')[j]}, res cold {b.get_stats('res')[j]}")
