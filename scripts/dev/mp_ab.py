"""A/B of the micro-panel forms (TUM_MP_LANES 0 / 1): N = 40 and N = 48, cold start: kernel times, iteration counts, difference of the iterates between the
two libraries and of each against the oracle on a sample (largest entry and where)."""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch  # noqa
from tum_control_amd import solver as sv
from tum_control_amd.workloads import nominal_batch
from oracle.oracle import OracleOcp
from tum_control_amd import config
libs = sys.argv[1:] or ['exp_libs/lib_tree_mp0.so', 'shipped']
B = 4096
for N in (40, 48):
    x0, yref = nominal_batch(B, N=N, seed=40 + N)
    res = {}
    for lib in libs:
        p = sv.LIB_PATH if lib == 'shipped' else os.path.abspath(lib)
        sv.load_library(p); sv._default_path = p
        s = sv.BatchedOcpSolver(N=N, batch=B); s.install_reference_ocp(); s.set_kernel("time-ipm"); s.set_x0(x0); s.set_yref_all(yref)
        ms, ipm = [], []
        for r in range(8):
            s.cold_start(); s.solve(); ms.append(s.last_kernel_ms()); ipm.append(1e3 * s.get_stats('time_ipm'))
        X, U = s.get_iterate()
        res[lib] = (X, U, s.get_stats('qp_iter').copy())
        print(f"{lib:28s} N {N}: solve {np.median(ms[2:]):.4f} ms, ipm_kernel {np.median(ipm[2:]):.4f} ms, qp_iter {res[lib][2].mean():.4f}, ok {(s.get_stats('status') == 0).mean():.4f}")
        del s
    a, b = res[libs[0]], res[libs[-1]]
    dU = np.abs(a[1] - b[1]); i = np.unravel_index(dU.argmax(), dU.shape)
    print(f"   between the two: iteration counts differ on {(a[2] != b[2]).sum()} of {B}; max |dU| {dU.max():.3e} at (instance, stage, input) {i}; max |dX| {np.abs(a[0] - b[0]).max():.3e}")
    m = config.MPC
    for lib in libs:
        worst = 0.0; where = None
        for j in range(0, B, B // 32):
            o = OracleOcp(N, 0.08, 3)
            o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
            o.cold_start(x0[j]); o.yref[:] = yref[j]; o.solve()
            d = np.abs(o.U - res[lib][1][j])
            if d.max() > worst:
                worst, where = d.max(), (j,) + tuple(int(v) for v in np.unravel_index(d.argmax(), d.shape)) + (o.qp_iter, int(res[lib][2][j]))
        print(f"   {lib:28s} vs oracle (32 instances): max |dU| {worst:.3e} at (instance, stage, input, oracle it, gpu it) {where}")
