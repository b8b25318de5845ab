#!/usr/bin/env python3
"""
Interior point iteration study (round 5, oracle first): does a WARM START of the interior point method from the previous QP's
multipliers, or Gondzio's MULTIPLE CENTRALITY CORRECTORS, save interior point WORK on the workloads that matter --

  cold   BASELINE configs[1]: 4096 perturbed cold starts, N = 40 (the headline batch; `--cold-n` instances of it)
  warm   the 52 complete logged acados closed loops, 285 948 warm-started real-time iterations, replayed per solve by
         tests/golden/replay_full_logs.py with its parity gate (the gate and its 30 exceptions must not get worse)

Everything is run with the CPU oracle (oracle/nmpc_oracle.c, switches in ipm_opts; all switches 0 = the method as shipped until round 4; since round 5 the warm start "warm5-mu1e-2" is the default). Work model, from
the measured phase split of ipm_kernel (profiles/r04_phase_cycles.txt): one factorisation (assembly + LDL') is half of a
predictor-corrector iteration, one back-solve with its row phases a quarter:  work = 0.5 x factorisations + 0.25 x back-solves
(a Mehrotra iteration = 1.0; an extra centrality corrector = 0.25). ADOPT only what saves >= 8 % of the work on both legs without
hurting the gate.  Output: profiles/r05_ipm_iterations.txt (this script's stdout).

Reference: the SNMPC solver is created with qp_solver_warm_start = 1 (Stochastic_NMPC/SNMPC_acados_settings.py:307); the nominal one
with HPIPM's cold start (NMPC_STM_acados_settings.py:230-240).
usage: ipm_iterations.py [--cold-n 1024] [--logs monteblanco:0,lvms:0,...] [--variants name,...] [--procs P]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

# name -> (warm variant, warm_mu, ncorr, dalpha)
VARIANTS = {
    "round-4":            (0, 0.0, 0, 0.0),
    "corr1":              (0, 0.0, 1, 0.1),
    "corr2":              (0, 0.0, 2, 0.1),
    "corr1-d0.3":         (0, 0.0, 1, 0.3),
    "corr2-d0.3":         (0, 0.0, 2, 0.3),
    "warm1-mu1e-2":       (1, 1e-2, 0, 0.0),
    "warm1-mu1e-3":       (1, 1e-3, 0, 0.0),
    "warm1-mu1e-4":       (1, 1e-4, 0, 0.0),
    "warm2-mu1e-2":       (2, 1e-2, 0, 0.0),
    "warm2-mu1e-3":       (2, 1e-3, 0, 0.0),
    "warm2-mu1e-4":       (2, 1e-4, 0, 0.0),
    "warm3-mu1e-3":       (3, 1e-3, 0, 0.0),
    "warm3-mu1e-4":       (3, 1e-4, 0, 0.0),
    "warm4-mu1e-3":       (4, 1e-3, 0, 0.0),
    "warm4-mu1e-4":       (4, 1e-4, 0, 0.0),
    "warm5-mu1e-2":       (5, 1e-2, 0, 0.0),
    "warm5-mu7e-3":       (5, 7e-3, 0, 0.0),
    "warm5-mu4e-3":       (5, 4e-3, 0, 0.0),
    "warm6-mu5e-3":       (6, 5e-3, 0, 0.0),
    "warm5-mu1e-3":       (5, 1e-3, 0, 0.0),
    "warm5-mu1e-4":       (5, 1e-4, 0, 0.0),
    "warm6-mu1e-3":       (6, 1e-3, 0, 0.0),
    "warm5-mu2e-2":       (5, 2e-2, 0, 0.0),
    "warm5-mu5e-3":       (5, 5e-3, 0, 0.0),
    "warm5-mu3e-3":       (5, 3e-3, 0, 0.0),
    "warm2-mu1e-3+corr1": (2, 1e-3, 1, 0.1),
    # termination tolerances (stat, ineq, comp) other than the shipped 1e-8 x 3: HPIPM's defaults are (recalled) res_g 1e-6, res_b / res_d / res_m 1e-8
    "tol-stat1e-6":       (0, 0.0, 0, 0.0, (1e-6, 1e-8, 1e-8)),
    "tol-comp1e-7":       (0, 0.0, 0, 0.0, (1e-8, 1e-8, 1e-7)),
    "tol-all1e-7":        (0, 0.0, 0, 0.0, (1e-7, 1e-7, 1e-7)),
    "tol-all1e-6":        (0, 0.0, 0, 0.0, (1e-6, 1e-6, 1e-6)),
    "warm5-mu1e-2+tol-stat1e-6": (5, 1e-2, 0, 0.0, (1e-6, 1e-8, 1e-8)),
    # primal start at the unconstrained minimiser (sixth entry: (mode, |q| threshold))
    "vstart":             (0, 0.0, 0, 0.0, None, (1, 0.0)),
    "vstart-q1":          (0, 0.0, 0, 0.0, None, (2, 1.0)),
    "vstart+warm5-mu1e-2": (5, 1e-2, 0, 0.0, None, (1, 0.0)),
    # separate primal / dual step lengths (seventh entry)
    "split":              (0, 0.0, 0, 0.0, None, (0, 0.0), 1),
    "split+warm5-mu1e-2": (5, 1e-2, 0, 0.0, None, (0, 0.0), 1),
    "dualclamp":          (0, 0.0, 0, 0.0, None, (0, 0.0), 2),
    "dualclamp+warm5-mu1e-2": (5, 1e-2, 0, 0.0, None, (0, 0.0), 2),
    "dualclamp+warm5-mu3e-3": (5, 3e-3, 0, 0.0, None, (0, 0.0), 2),
    "split+warm5-mu3e-3": (5, 3e-3, 0, 0.0, None, (0, 0.0), 1),
    "split+warm5-mu5e-3": (5, 5e-3, 0, 0.0, None, (0, 0.0), 1),
    "split+warm5-mu7e-3": (5, 7e-3, 0, 0.0, None, (0, 0.0), 1),
    "split+warm5-mu2e-2": (5, 2e-2, 0, 0.0, None, (0, 0.0), 1),
    "split+warm6-mu5e-3": (6, 5e-3, 0, 0.0, None, (0, 0.0), 1),
    "split+warm2-mu5e-3": (2, 5e-3, 0, 0.0, None, (0, 0.0), 1),
    # the corrector pass skipped when the predictor asks for (next to) no centring (eighth entry: (sigma threshold, step-to-boundary threshold))
    "skip-s1e-3-a0.9":    (0, 0.0, 0, 0.0, None, (0, 0.0), 0, (1e-3, 0.9)),
    "skip-s1e-2-a0.8":    (0, 0.0, 0, 0.0, None, (0, 0.0), 0, (1e-2, 0.8)),
    "skip-s1e-4-a0.95":   (0, 0.0, 0, 0.0, None, (0, 0.0), 0, (1e-4, 0.95)),
    "skip-s1e-1-a0.5":    (0, 0.0, 0, 0.0, None, (0, 0.0), 0, (1e-1, 0.5)),
    "warm5-mu1e-2+skip-s1e-3-a0.9": (5, 1e-2, 0, 0.0, None, (0, 0.0), 0, (1e-3, 0.9)),
    "warm5-mu1e-2+skip-s1e-2-a0.8": (5, 1e-2, 0, 0.0, None, (0, 0.0), 0, (1e-2, 0.8)),
}


def work_of(w):
    return 0.5 * w[1] + 0.25 * w[2]


def cold_leg(n, var):
    from oracle.oracle import OracleOcp, global_work
    from tum_control_amd import config
    from tum_control_amd.workloads import nominal_batch
    x0, yref = nominal_batch(4096, N=40)
    x0, yref = x0[:n], yref[:n]
    m = config.MPC
    o = OracleOcp(40, 0.08, 3)
    o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
    o.set_ipm_experiment(*var[:4])
    if len(var) > 4 and var[4]:
        o.ipm_tol[:] = var[4]
    if len(var) > 5:
        o.set_ipm_vstart(*var[5])
    if len(var) > 6:
        o.set_ipm_split(var[6])
    if len(var) > 7:
        o.set_ipm_skip(*var[7])
    global_work(reset=True)
    u0, X1, st = o.solve_batch_cold(x0, yref, max(1, len(os.sched_getaffinity(0))))
    w = global_work(reset=True)
    return dict(it=float(st[:, 1].mean()), itmax=int(st[:, 1].max()), ok=float((st[:, 2] == 0).mean()), work=work_of(w) / n, u0=u0, x1=X1)


def warm_leg(logs, var, procs):
    import replay_full_logs as R
    os.environ["REPLAY_IPM"] = ",".join(str(v) for v in var[:4])
    os.environ.pop("REPLAY_TOL", None)
    os.environ.pop("REPLAY_VSTART", None)
    if len(var) > 4 and var[4]:
        os.environ["REPLAY_TOL"] = ",".join(str(v) for v in var[4])
    os.environ.pop("REPLAY_SPLIT", None)
    if len(var) > 5:
        os.environ["REPLAY_VSTART"] = ",".join(str(v) for v in var[5])
    if len(var) > 6:
        os.environ["REPLAY_SPLIT"] = str(var[6])
    os.environ.pop("REPLAY_SKIP", None)
    if len(var) > 7:
        os.environ["REPLAY_SKIP"] = ",".join(str(v) for v in var[7])
    rep = R.run(logs, procs)
    n = sum(r["n"] for r in rep)
    w = np.sum([r["work"] for r in rep], axis=0)
    # (the tightened re-solves of candidate exceptions run through the same counters: a handful per loop)
    out = dict(it=sum(r["mean_qp_iter"] * r["n"] for r in rep) / n, itmax=max(r["max_qp_iter"] for r in rep), work=work_of(w) / w[0],
               n_exc=sum(len(r["exceptions"]) for r in rep), worst=max(r["worst_comparable"] for r in rep),
               n_above_tol=sum(r["n_above_tol"] for r in rep), n_above_1e6=sum(r["n_above_1e6"] for r in rep), n=n)
    try:
        out["gate"] = R.gate(rep) if logs is None else "(subset: gate not evaluated)"
    except AssertionError as e:
        out["gate"] = "GATE FAILS: " + str(e)[:200]
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    cold_n = int(a[a.index("--cold-n") + 1]) if "--cold-n" in a else 1024
    procs = int(a[a.index("--procs") + 1]) if "--procs" in a else None
    logs = None
    if "--logs" in a:
        logs = [(s.split(":")[0], int(s.split(":")[1])) for s in a[a.index("--logs") + 1].split(",")]
    names = a[a.index("--variants") + 1].split(",") if "--variants" in a else list(VARIANTS)
    base_c = base_w = None
    print(f"# interior point iteration study: cold = {cold_n} instances of BASELINE configs[1] (N = 40); warm = "
          f"{'all 52 logged loops' if logs is None else logs} (N = 38); work = 0.5 x factorisations + 0.25 x back-solves per QP", flush=True)
    for nm in names:
        var = VARIANTS[nm]
        t = time.time()
        c = cold_leg(cold_n, var) if (var[0] == 0 or len(var) > 5) else None          # (a cold start has no previous QP: the warm variants do not change it)
        w = warm_leg(logs, var, procs)
        if nm == "round-4":
            base_c, base_w = c, w
        line = f"{nm:20s}"
        if c is not None:
            dev = np.abs(c["u0"] - base_c["u0"]).max() if base_c is not None else 0.0
            line += f" cold: it {c['it']:.3f} (max {c['itmax']}) work {c['work']:.3f}" + (f" ({100 * (c['work'] / base_c['work'] - 1):+.1f} %)" if base_c else "") + f" ok {c['ok']:.4f} |du0| vs round-4 {dev:.1e};"
        else:
            line += " cold: (= round-4);"
        line += (f" warm: it {w['it']:.3f} (max {w['itmax']}) work {w['work']:.3f}" + (f" ({100 * (w['work'] / base_w['work'] - 1):+.1f} %)" if base_w else "") +
                 f" above 1e-4: {w['n_above_tol']} above 1e-6: {w['n_above_1e6']} worst {w['worst']:.1e} exceptions {w['n_exc']}; {w['gate']}  [{time.time() - t:.0f} s]")
        print(line, flush=True)
