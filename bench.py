#!/usr/bin/env python3
"""
bench.py -- SQP-RTI OCP solves/sec on synthetic batches (BASELINE.json metric).

  python bench.py [--config {2,3,4,5}] --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; the batch is sharded by scenario GROUP, every rank solves
   its own share, and the result slab is gathered to rank 0 over RCCL each step: one collective per step.)

A "step" = one pass of the hot path over one batch, inputs resident in HBM before the timed region:
  config 2 (default; BASELINE configs[1], the configuration the metric is quoted on): cold start + one SQP real-time
           iteration for 4096 nominal OCPs per GPU (perturbed x0, Monteblanco, N=40, one wavefront per OCP);
  config 3 (configs[2]): 16384 sigma-point scenarios per GPU (1024 poses x (nominal + 15)), cold start + SQP-RTI + PCE
           mean / variance of x_1 per scenario group (K6) on the device;
  config 4 (configs[3]): 16384 Monte-Carlo scenarios per GPU (131072 over 8 GPUs), LVMS, cold start + SQP-RTI;
  config 5 (configs[4]): 4096 R2NMPC instances per GPU (32768 over 8), Modena: nominal bounds, cold start, SQP-RTI,
           covariance back-off (K7), SQP-RTI with the tightened bounds (2 solves per step).
Prints ONE JSON line on rank 0. At N=1 the line also carries the natural-order and the fresh-batch figures of the schedule
 and the CPU baseline (the oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = matrix peak (spec); v_mfma_f64_16x16x4 probe: 70.2 measured


def algorithmic_bytes(N, warm=False, per_instance_yref=True):
    """FP64 bytes one solve must move. cold/shared-yref 3344 B, warm/per-instance 8560 B at N=40."""
    X, U, yref = (N + 1) * 8, N * 2, N * 6 + 4
    rd = 8 + (yref if per_instance_yref else 0) + ((X + U) if warm else 0)
    wr = X + U + 2
    return 8 * (rd + wr)


def algorithmic_flops(N, nsub, qp_iter):
    """SURVEY.md 8(d) formulas (FMA = 2 FLOP)."""
    nx, nu = 8, 2
    nv, ng = nu * N, 2 * N
    dyn = 4 * nsub * N * (250 + 350 + 2 * nx * nx * (nx + nu))
    condG = 128 * N * (N + 1)
    condH = 16 * N ** 3 / 3.0
    condC = 2 * 2 * nx * nu * N * (N + 1) / 2.0
    ipm = ng * nv * nv + nv ** 3 / 3.0 + 6 * nv * nv + 4 * ng * nv
    return dyn + condG + condH + condC + qp_iter * ipm


def usable_cores():
    """Host threads this process may really use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(N, x0, yref, cfg, budget_s=15.0):
    """The oracle (CPU restatement, kind 'port') timed on the host cores on a bounded sample; rebuilt -march=native for this
    host when a compiler is present."""
    from oracle import oracle as _oracle
    build = _oracle.use_native()
    from oracle.oracle import OracleOcp
    mpc = cfg["mpc"]
    o = OracleOcp(N, 0.08, 3)
    o.set_weights(mpc["q_lon"], mpc["q_yaw"], mpc["q_vel"], mpc["r_jerk"], mpc["r_steering_rate"],
                  mpc["L1_pen"], mpc["L2_pen"], scale=0.01)
    o.zl[:] = mpc["L1_pen"]; o.zu[:] = mpc["L1_pen"]; o.Zl[:] = mpc["L2_pen"]; o.Zu[:] = mpc["L2_pen"]
    cores = usable_cores()
    # calibrate on a few solves, then size the sample for ~budget_s of CPU work
    t0 = time.perf_counter(); o.solve_batch_cold(x0[:8], yref[:8], 1); t1 = time.perf_counter()
    per = max((t1 - t0) / 8, 1e-5)
    ns1 = int(min(len(x0), max(16, 0.3 * budget_s / per)))
    t0 = time.perf_counter(); u1, _, st1 = o.solve_batch_cold(x0[:ns1], yref[:ns1], 1); t1 = time.perf_counter()
    single = ns1 / (t1 - t0)
    nsm = int(min(len(x0), max(64, 0.7 * budget_s * single * cores * 0.7)))
    reps, t0 = 0, time.perf_counter()
    while True:                      # repeat the sample until >= 1.5 s of wall time (steady state under a CPU quota)
        um, _, stm = o.solve_batch_cold(x0[:nsm], yref[:nsm], cores); reps += 1
        t1 = time.perf_counter()
        if t1 - t0 >= 1.5:
            break
    multi = reps * nsm / (t1 - t0)
    return dict(value=multi, unit="OCP solves/s", cores=cores, kind="port", cpu_model=cpu_model(), build=build,
                sample=f"{reps} x {nsm} cold-start solves of the same batch on {cores} threads (OpenMP over instances; cores = min(affinity, cgroup cpu quota)); "
                       f"single thread: {single:.1f} solves/s on {ns1} solves; mean qp_iter {float(stm[:,1].mean()):.1f}",
                single_thread=single), um[:ns1] if nsm >= ns1 else u1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5), help="BASELINE.json configs[config-1]")
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default: the config's per-GPU share)")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-schedule-legs", action="store_true", help="skip the natural-order and fresh-batch legs (N=1 only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("RANK") is not None     # launched by torch.distributed.run
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the solver has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    from tum_control_amd import config as _cfg
    from tum_control_amd import sharding
    from tum_control_amd.solver import BatchedOcpSolver
    from tum_control_amd.workloads import CONFIGS, config_groups

    N, cid = args.horizon, args.config
    C = CONFIGS[cid]
    gsz = C["group"]
    # weak scaling: every rank owns the config's per-GPU share, a whole number of scenario groups (contiguous block of the
    # global group range: a group never straddles two ranks)
    groups_per_gpu = C["groups_per_gpu"] if args.batch is None else max(1, args.batch // gsz)
    groups_total = world * groups_per_gpu
    g_lo, g_hi, b_lo, b_hi = sharding.shard_groups(groups_total, gsz, world, rank)
    B = b_hi - b_lo
    x0, yref, _ = config_groups(cid, g_lo, g_hi, groups_total, N=N, dt=0.08)
    assert len(x0) == B
    P = g_hi - g_lo

    s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B, device=local_rank, store_qp_in=(cid == 5))
    s.install_reference_ocp()
    s.set_x0(x0); s.set_yref_all(yref)
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    nmom = 0
    if cid == 3:      # PCE matrix of the 15 Hammersley sigma points (10 terms): K6 runs inside the step
        from tum_control_amd.snmpc import alpha_generation, hammersley_normal, pce_matrix
        s.pce_attach(pce_matrix(hammersley_normal(15, 3), alpha_generation(3, 2)))
        nmom = 16
    if cid == 5:      # covariance back-off attached to every solve (K7), nominal bounds restored at the start of a step
        from tum_control_amd.r2nmpc import r2_setup
        m, veh = s.cfg["mpc"], s.cfg["veh"]
        S0, BWB = r2_setup(m["stds"], 0.08)
        s.r2_attach(S0, BWB, int(m["uncertainty_propagation_horizon"]), veh["delta_f_min"], veh["delta_f_max"], 1.0)
        s.bounds_snapshot()

    # result slab gathered to rank 0 each step: (u0[2], cost, status, qp_iter) as 5 doubles per instance, for config 3
    # followed by the PCE mean / variance of x_1 of every scenario group (16 doubles per group): ONE flat buffer, packed
    # on the device by the library, moved with ONE rooted gather
    slab = torch.zeros(B * 5 + P * nmom, dtype=torch.float64, device=dev)
    mom_ptr = slab.data_ptr() + 8 * B * 5
    gather = sharding.ResultGatherer(world, rank, 1, dev, nf=slab.numel(), ni=1) if distributed else None
    spp = C["solves_per_step"]

    def step(ev=None):
        if cid == 5:
            s.bounds_restore()
        s.cold_start()
        for j in range(spp):
            if ev is not None:
                ev[2 * j].record()
            s.solve_async()
            if ev is not None:
                ev[2 * j + 1].record()
        if cid == 3:
            s.pce_moments_device("x", 1, mom_ptr, mom_ptr + 8 * P * 8)
        if gather is not None:
            s.get_device("summary", slab.data_ptr())
            gather.gather(slab.view(1, -1))

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def new_events(n):
        return [[torch.cuda.Event(enable_timing=True) for _ in range(2 * spp)] for _ in range(n)]

    def kernel_ms(evs):
        return float(np.mean([e[2 * j].elapsed_time(e[2 * j + 1]) for e in evs for j in range(spp)]))

    for _ in range(args.warmup):
        step()
    barrier()
    evs = new_events(args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(evs[i])
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = kernel_ms(evs)

    # correctness of what was timed: statuses, and a parity spot check against the oracle on rank 0
    st = s.get_stats("status"); it = s.get_stats("qp_iter")
    X, U = s.get_iterate()
    # which kernels ran: the four-kernel pipeline (the default), whose dominant
    # kernel -- the interior point kernel -- is timed on its own (library events around it) over five more steps
    ipm_ms = None
    try:
        tm = []
        for _ in range(5):
            step(); torch.cuda.synchronize(); tm.append(1e3 * s.get_stats("time_ipm"))
        ipm_ms = float(np.median(tm))
    except Exception:
        pass

    # The schedule legs (N = 1): (a) the same batch with the instances dispatched in natural order; (b) FRESH batches: four
    # differently seeded batches of the same workload resident in HBM, rotated every step, so that the longest-first order
    # comes from a DIFFERENT batch's iteration counts (stale history = what a caller with a new batch every step sees).
    nat_ms = fresh_ms = fresh_value = None
    if world == 1 and not args.no_schedule_legs:
        s.set_schedule(False)
        nat = new_events(5)
        for e in nat:
            step(e)
        torch.cuda.synchronize()
        nat_ms = float(np.median([e[2 * j].elapsed_time(e[2 * j + 1]) for e in nat for j in range(spp)]))
        s.set_schedule(True)
        nb = 4
        shift = groups_per_gpu
        dx0, dyr = [], []
        for k in range(nb):      # batch k = the group range shifted by k * (groups per GPU): other poses, other random streams
            gx, gy, _ = config_groups(cid, g_lo + (k + 1) * shift, g_hi + (k + 1) * shift, groups_total * (nb + 1), N=N, dt=0.08)
            dx0.append(torch.from_numpy(np.ascontiguousarray(gx)).to(dev)); dyr.append(torch.from_numpy(np.ascontiguousarray(gy)).to(dev))

        def fresh_step(k, ev=None):
            s.put_device("x0", dx0[k % nb].data_ptr()); s.put_device("yref", dyr[k % nb].data_ptr())
            step(ev)

        for k in range(nb):
            fresh_step(k)
        torch.cuda.synchronize()
        fev = new_events(args.steps)
        t0 = time.perf_counter()
        for k in range(args.steps):
            fresh_step(k, fev[k])
        torch.cuda.synchronize()
        fresh_elapsed = time.perf_counter() - t0
        fresh_ms = kernel_ms(fev)
        fresh_value = B * spp * args.steps / fresh_elapsed
        fresh_ok = float((s.get_stats("status") == 0).mean())

    if rank == 0:
        total = world * B * spp * args.steps
        value = total / elapsed
        mean_it = float(it.mean())
        shared_yref = gsz > 1
        flops = algorithmic_flops(N, 3, mean_it) * B
        abytes = algorithmic_bytes(N, warm=(cid == 5), per_instance_yref=not shared_yref) * B
        # HBM bytes per launch from the PMC counters cannot be collected from inside this process; they come
        # from the committed rocprofv3 --pmc passes of this same command (profiles/*_traffic.json), only when
        # the workload matches what was profiled.
        traffic, traffic_src = None, None
        try:
            tj = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))[-1]
            tr = json.load(open(os.path.join(ROOT, "profiles", tj)))
            if tr.get("batch") == B and tr.get("N") == N and tr.get("config", 2) == cid:
                traffic, traffic_src = tr["traffic_bytes_per_launch"], "profiles/" + tj
        except Exception:
            pass
        ach_tf = flops / (kern_ms * 1e-3) / 1e12
        ach_gb = abytes / (kern_ms * 1e-3) / 1e9
        nv_, ng_ = 2 * N, 2 * N
        ipm_flops = mean_it * (ng_ * nv_ * nv_ + nv_ ** 3 / 3.0 + 6 * nv_ * nv_ + 4 * ng_ * nv_) * B
        if ipm_ms:
            kname = "pipeline of lin_kernel + cond_kernel + ipm_kernel + expand_kernel (kernel_ms = device time of one solve)"
            dominant = {"kernel": "ipm_kernel", "kernel_ms": ipm_ms, "flops_per_launch": ipm_flops,
                        "achieved": ipm_flops / (ipm_ms * 1e-3) / 1e12, "frac": ipm_flops / (ipm_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
        else:
            kname, dominant = "nmpc_rti_kernel", None
        out = {
            "metric": "SQP-RTI OCP solves/sec (batch), N=40 single-track Pacejka",
            "value": value, "unit": "OCP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{cid - 1}]: {C['name']}, {C['track']} reftraj, cold-start SQP-RTI, one wavefront per OCP",
                       "N": N, "nx": 8, "nu": 2, "nsub": 3, "batch_per_gpu": B, "global_batch": world * B,
                       "scenario_group": gsz, "groups_per_gpu": P, "solves_per_step": spp,
                       "parallelism": f"scenario groups sharded x{world} (a group never straddles ranks), RCCL gather of "
                                      f"{slab.numel() * 8} B per rank (u0, cost, status, qp_iter"
                                      + (", PCE mean/var of x_1 per group" if nmom else "") + "), one collective per step",
                       "schedule": "workgroups take instances longest-first by the previous solve's IPM iteration count "
                                   "(tum_ocp_set_schedule): exact history when the same batch is solved again (`value`), "
                                   "stale when every step brings a new batch (`value_fresh_batch`)",
                       "kernel_ms_natural_order": nat_ms,
                       "solves_per_s_per_gpu_natural_order": (B / nat_ms * 1e3) if nat_ms else None,
                       "kernel_ms_fresh_batch": fresh_ms},
            "value_fresh_batch": fresh_value,
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms": kern_ms, "dominant": dominant, "mean_qp_iter": mean_it,
                         "flops_per_solve": flops / B,
                         "hbm": {"achieved": ach_gb, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_gb / HBM_PEAK_GBPS,
                                 "bytes_per_solve": abytes / B}},
            "status_ok_frac": float((st == 0).mean()),
        }
        if fresh_value is not None:
            out["status_ok_frac_fresh_batch"] = fresh_ok
        if not args.no_cpu_baseline and world == 1:
            cb, u_ref = cpu_baseline(N, x0, yref, s.cfg)
            out["cpu_baseline"] = cb
            if cid != 5:      # (config 5's iterate is the SECOND solve; the oracle sample is the cold-start solve)
                n = len(u_ref)
                err = np.abs(U[:n, 0] - u_ref).max(axis=1)
                out["parity_vs_oracle_max_abs_u0"] = float(err.max())
                out["parity_vs_oracle_frac_within_1e-6"] = float((err < 1e-6).mean())
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
