#!/usr/bin/env python3
"""
bench.py -- SQP-RTI OCP solves/sec on synthetic batches (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; the batch is sharded, every rank
   solves its own 4096 instances, and the results are gathered to rank 0 over RCCL each step.)

A "step" = one pass of the hot path over one batch: cold start + one SQP real-time iteration for
every instance (BASELINE config 2: nominal NMPC, perturbed x0, Monteblanco, N=40, one wavefront
per OCP). Inputs are resident in HBM before the timed region. Prints ONE JSON line on rank 0.
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


def cpu_baseline(N, x0, yref, cfg, budget_s=15.0):
    """The oracle (CPU restatement, kind 'port') timed on the host cores on a bounded sample."""
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
    return dict(value=multi, unit="OCP solves/s", cores=cores, kind="port",
                sample=f"{reps} x {nsm} cold-start solves of the same batch on {cores} threads (OpenMP over instances; cores = min(affinity, cgroup cpu quota)); "
                       f"single thread: {single:.1f} solves/s on {ns1} solves; mean qp_iter {float(stm[:,1].mean()):.1f}",
                single_thread=single), um[:ns1] if nsm >= ns1 else u1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compare-schedules", action="store_true",
                    help="after the timed region, also time 5 launches with the instances in natural order")
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

    from tum_control_amd.solver import BatchedOcpSolver
    from tum_control_amd.workloads import nominal_batch
    from tum_control_amd import sharding

    N, B = args.horizon, args.batch
    # weak scaling: every rank owns `B` instances of the global batch world*B (contiguous block)
    x0, yref = nominal_batch(B, N=N, dt=0.08, track_name="monteblanco", stride=37, seed=1234 + rank, offset=rank * B)
    s = BatchedOcpSolver(N=N, dt=0.08, nsub=3, batch=B, device=local_rank)
    s.install_reference_ocp()
    s.set_x0(x0); s.set_yref_all(yref)
    s.set_stream(torch.cuda.current_stream().cuda_stream)

    # result slab gathered to rank 0 each step: (u0[2], cost, status, qp_iter) as 5 doubles per instance, packed on the
    # device by the library (one kernel) and moved with ONE rooted gather
    res = torch.zeros((B, 5), dtype=torch.float64, device=dev)
    gather = sharding.ResultGatherer(world, rank, B, dev, nf=5, ni=1) if distributed else None

    def step(ev=None):
        s.cold_start()
        if ev is not None:
            ev[0].record()
        s.solve_async()
        if ev is not None:
            ev[1].record()
        if gather is not None:
            s.get_device("summary", res.data_ptr())
            gather.gather(res)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
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
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    # on request: the same kernel with the instances dispatched in natural order (see config.schedule)
    nat_ms = None
    if args.compare_schedules:
        s.set_schedule(False)
        nat = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.cold_start(); e0.record(); s.solve_async(); e1.record(); torch.cuda.synchronize()
            nat.append(e0.elapsed_time(e1))
        s.set_schedule(True)
        nat_ms = float(np.median(nat))

    # correctness of what was timed: statuses, and a parity spot check against the oracle on rank 0
    st = s.get_stats("status"); it = s.get_stats("qp_iter")
    X, U = s.get_iterate()
    out = None
    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        mean_it = float(it.mean())
        flops = algorithmic_flops(N, 3, mean_it) * B
        abytes = algorithmic_bytes(N, warm=False, per_instance_yref=True) * B
        # HBM bytes per launch from the PMC counters cannot be collected from inside this process; they come
        # from the committed rocprofv3 --pmc passes of this same command (profiles/*_traffic.json), only when
        # the workload matches what was profiled.
        traffic, traffic_src = None, None
        try:
            tj = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))[-1]
            tr = json.load(open(os.path.join(ROOT, "profiles", tj)))
            if tr.get("batch") == B and tr.get("N") == N:
                traffic, traffic_src = tr["traffic_bytes_per_launch"], "profiles/" + tj
        except Exception:
            pass
        ach_tf = flops / (kern_ms * 1e-3) / 1e12
        ach_gb = abytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "SQP-RTI OCP solves/sec (batch), N=40 single-track Pacejka",
            "value": value, "unit": "OCP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: nominal NMPC, perturbed x0, Monteblanco reftraj, cold-start SQP-RTI, "
                                   "one wavefront per OCP", "N": N, "nx": 8, "nu": 2, "nsub": 3, "batch_per_gpu": B,
                       "global_batch": world * B, "parallelism": f"instances sharded x{world}, RCCL gather of (u0,cost,status,qp_iter), 40 B per instance, one collective per step",
                       "schedule": "workgroups take instances longest-first by the previous solve's IPM iteration count "
                                   "(tum_ocp_set_schedule); natural order: --compare-schedules, profiles/*_schedules.json",
                       "kernel_ms_natural_order": nat_ms,
                       "solves_per_s_per_gpu_natural_order": (B / nat_ms * 1e3) if nat_ms else None},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "nmpc_rti_kernel", "kernel_ms": kern_ms, "mean_qp_iter": mean_it,
                         "flops_per_solve": flops / B,
                         "hbm": {"achieved": ach_gb, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_gb / HBM_PEAK_GBPS,
                                 "bytes_per_solve": abytes / B}},
            "status_ok_frac": float((st == 0).mean()),
        }
        if not args.no_cpu_baseline and world == 1:
            cb, u_ref = cpu_baseline(N, x0, yref, s.cfg)
            out["cpu_baseline"] = cb
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
