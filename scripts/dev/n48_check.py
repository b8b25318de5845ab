import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from tum_control_amd import solver as sv
from tum_control_amd.workloads import nominal_batch
for lib in sys.argv[1:]:
    p = sv.LIB_PATH if lib == 'shipped' else os.path.abspath(lib)
    sv.load_library(p); sv._default_path = p
    for N in (44, 48):
        x0, yref = nominal_batch(4096, N=N)
        s = sv.BatchedOcpSolver(N=N, batch=4096); s.install_reference_ocp(); s.set_x0(x0); s.set_yref_all(yref)
        ms = []
        for r in range(8):
            s.cold_start(); s.solve(); ms.append(s.last_kernel_ms())
        print(lib, 'N', N, 'kernel ms %.3f' % np.median(ms[2:]), 'ipm %.3f' % (1e3 * s.get_stats('time_ipm')), 'qp_iter %.2f' % s.get_stats('qp_iter').mean(), 'ok %.4f' % (s.get_stats('status') == 0).mean())
        del s
