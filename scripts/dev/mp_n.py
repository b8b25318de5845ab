import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch  # noqa
from tum_control_amd import solver as sv
from tum_control_amd.workloads import nominal_batch
B = 1024
for N in (41, 44, 45, 46, 47, 48):
    x0, yref = nominal_batch(B, N=N, seed=40 + N)
    res = {}
    for lib in ('exp_libs/lib_tree_mp0.so', os.environ.get('MP_B', 'shipped')):
        p = sv.LIB_PATH if lib == 'shipped' else os.path.abspath(lib)
        sv.load_library(p); sv._default_path = p
        s = sv.BatchedOcpSolver(N=N, batch=B); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref)
        s.cold_start(); s.solve()
        res[lib] = s.get_iterate() + (s.get_stats('qp_iter').copy(),)
        del s
    a, b = res['exp_libs/lib_tree_mp0.so'], res[os.environ.get('MP_B', 'shipped')]
    dU = np.abs(a[1] - b[1])
    print(N, 'max dU per stage (last 4 stages):', dU.max(axis=(0))[-4:].tolist(), 'max elsewhere', dU[:, :-4].max(), 'iter diff', (a[2] != b[2]).sum())
