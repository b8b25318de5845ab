"""development aid: the same solve with the tile count the horizon needs and with a larger instantiation (TUM_FORCE_TILES): the padding must not matter"""
import sys, os, subprocess, numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch  # noqa
    from tum_control_amd.solver import BatchedOcpSolver
    from tum_control_amd.workloads import nominal_batch
    N, B = int(sys.argv[2]), 64
    x0, yref = nominal_batch(B, N=N, seed=40 + N)
    s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B); s.install_reference_ocp()
    s.set_x0(x0); s.set_yref_all(yref); s.cold_start(); st = s.solve()
    X, U = s.get_iterate()
    np.savez(sys.argv[3], X=X, U=U, it=s.get_stats("qp_iter"), st=s.get_stats("status"))
else:
    for N in (30, 40, 45, 48):
        out = {}
        for f in (0, 6, 7):
            if f and f <= (6 if N > 40 else 5): continue
            env = dict(os.environ, TUM_FORCE_TILES=str(f))
            fn = f"/tmp/tiles_{N}_{f}.npz"
            subprocess.run([sys.executable, __file__, "child", str(N), fn], env=env, check=True, stderr=subprocess.DEVNULL)
            out[f] = np.load(fn)
        for f in out:
            if f: print(f"N {N}: forced {f} tiles vs native: max |dU| {np.abs(out[f]['U'] - out[0]['U']).max():.2e}, status0 {np.mean(out[f]['st'] == 0):.3f}, qp_iter {out[f]['it'].mean():.2f} vs {out[0]['it'].mean():.2f}")
