#!/usr/bin/env python3
"""
bench.py -- SQP-RTI OCP solves/sec on synthetic batches (BASELINE.json metric).

  python bench.py [--config {2,3,4,5}] [--scaling {weak,strong}] --gpus N --steps K --warmup W
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

`value` is the FRESH-BATCH throughput: NB differently seeded batches of the workload are resident in HBM and rotated, one per
step, so no step sees a batch the solver has just solved (what a caller with new data every step gets). The steps are dealt
to --streams S capsules in turn (default 3 -- four win 2 % in a long stream of batches and lose on this program's short legs, profiles/r06_stream_counts.txt; tum-control_amd/streaming.py), each with its own buffers on its own HIP stream: a
step is still one complete pass over one batch, but the GPU starts on the next batch while the last wavefronts of the
previous one finish (a batch is only four rounds of resident wavefronts: run one at a time, a fifth of the chip idles in
every batch's tail). At N = 1 the line also carries the same loop on ONE capsule / ONE stream (`value_single_stream`: the
roofline block and the per-kernel times are taken from that leg, where a launch has the chip to itself), the repeated-batch
figure (the same batch every step: the longest-first dispatch then has exact history, `value_repeated_batch`) and the
natural-order figure (`config.kernel_ms_natural_order`).

--scaling weak (default): every rank owns the config's per-GPU share (global batch = N x share).
--scaling strong: a fixed global batch (--global-batch, default 8 x the per-GPU share = the BASELINE multi-GPU size of
                  configs 4 / 5) is cut into N group-aligned shards.

Prints ONE JSON line on rank 0. At N = 1 the line also carries the CPU baseline (the oracle on the host cores).
The control flow below (`run`) is backend-agnostic: tests/test_host_logic.py drives it on CPU with gloo, world size 2 and a
stand-in solver.
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
NB_FRESH = 4                 # resident batches rotated by the timed loop (one more when that shares a factor with the stream count)


def algorithmic_bytes(N, warm=False, per_instance_yref=True):
    """FP64 bytes one solve must move. cold/shared-yref 3344 B, warm/per-instance 8560 B at N=40."""
    X, U, yref = (N + 1) * 8, N * 2, N * 6 + 4
    rd = 8 + (yref if per_instance_yref else 0) + ((X + U) if warm else 0)
    wr = X + U + 2
    return 8 * (rd + wr)


def ipm_flops_per_iter(N):
    nv, ng = 2 * N, 2 * N
    return ng * nv * nv + nv ** 3 / 3.0 + 6 * nv * nv + 4 * ng * nv


def algorithmic_flops(N, nsub, qp_iter):
    """SURVEY.md 8(d) formulas (FMA = 2 FLOP)."""
    nx, nu = 8, 2
    dyn = 4 * nsub * N * (250 + 350 + 2 * nx * nx * (nx + nu))
    condG = 128 * N * (N + 1)
    condH = 16 * N ** 3 / 3.0
    condC = 2 * 2 * nx * nu * N * (N + 1) / 2.0
    return dyn + condG + condH + condC + qp_iter * ipm_flops_per_iter(N)


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5), help="BASELINE.json configs[config-1]")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="--scaling strong: instances of the whole job (default: 8 x the config's per-GPU share)")
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (weak scaling; default: the config's per-GPU share)")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--streams", type=int, default=3,
                    help="capsules (each on its own HIP stream) the steps are dealt to in turn: step k runs on capsule k mod S, so the "
                         "tail of one batch's interior point kernel runs beside the head of the next batch (1 = one capsule, one stream)")
    ap.add_argument("--same-batch", action="store_true",
                    help="solve the SAME batch every step (batch 0; the longest-first dispatch then has exact history) instead of "
                         "rotating fresh batches: round 2's headline, kept for profiling the repeated-batch leg on its own")
    ap.add_argument("--gather-iterate", action="store_true",
                    help="the rooted gather carries the whole iterate as well (X, U: (N+1)*8 + N*2 doubles per instance behind the 40 B summary; "
                         "SURVEY 8(e): 3264 B per instance at N = 40, 53 MB per GPU on config 4)")
    ap.add_argument("--bind-inputs", action="store_true",
                    help="let the capsule read every step's resident batch (x0, yref) in place (tum_ocp_bind_device) instead of copying it "
                         "device-to-device into the capsule's own arrays inside the timed region (tum_ocp_put_device, the default and what "
                         "rounds 1-5 report; measured difference 4.037 vs 4.032 M solves/s, profiles/r05_bind_inputs.txt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the host-visible legs (N=1 only)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the bounded legs of configs 3 / 4 / 5 and the coupled SNMPC OCP that the default N=1 line carries as `other_configs`")
    ap.add_argument("--no-schedule-legs", action="store_true",
                    help="skip the repeated-batch, natural-order and N = 38 legs (N=1 only)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# Timing primitives: HIP events on the current torch stream on a GPU, the host clock on CPU (the gloo test).
class _Clock:
    def __init__(self, torch, cuda):
        self.torch, self.cuda = torch, cuda

    def mark(self):
        if self.cuda:
            e = self.torch.cuda.Event(enable_timing=True); e.record(); return e
        return time.perf_counter()

    def ms(self, a, b):
        return a.elapsed_time(b) if self.cuda else 1e3 * (b - a)

    def sync(self):
        if self.cuda:
            self.torch.cuda.synchronize()


class Job:
    """One rank's share of the benchmark job: solver, resident batches, the step (cold start, solve(s), reduction, gather) and
    the timed loop with its barrier / max-over-ranks protocol. Nothing here depends on the backend: `solver_factory`, `dev`
    and the process group decide whether this is RCCL on GPUs or the CPU stand-in of the tests."""

    def __init__(self, args, torch, dist, dev, world, rank, local_rank, solver_factory, workload=None, n_slots=None):
        from tum_control_amd import sharding
        from tum_control_amd.streaming import SolverRing
        from tum_control_amd.workloads import CONFIGS, config_groups
        self.args, self.torch, self.dist, self.dev = args, torch, dist, dev
        self.world, self.rank = world, rank
        self.distributed = dist is not None and dist.is_initialized()
        self.cuda = dev.type == "cuda"
        self.clock = _Clock(torch, self.cuda)
        N, cid = args.horizon, args.config
        self.N, self.cid = N, cid
        C = self.C = CONFIGS[cid]
        gsz = self.gsz = C["group"]
        share = C["groups_per_gpu"] if args.batch is None else max(1, args.batch // gsz)
        if args.scaling == "strong":
            gb = args.global_batch if args.global_batch is not None else 8 * C["groups_per_gpu"] * gsz
            if gb % gsz:
                raise SystemExit(f"--global-batch must be a multiple of the scenario group size {gsz}")
            groups_total = gb // gsz
            if groups_total < world:
                raise SystemExit("--global-batch: fewer scenario groups than ranks")
        else:
            groups_total = world * share
        self.groups_total = groups_total
        # contiguous block of the global group range: a group never straddles two ranks
        self.g_lo, self.g_hi, b_lo, b_hi = sharding.shard_groups(groups_total, gsz, world, rank)
        B = self.B = b_hi - b_lo
        P = self.P = self.g_hi - self.g_lo
        self.global_batch = groups_total * gsz
        gen = workload or (lambda k: config_groups(cid, self.g_lo, self.g_hi, groups_total, N=N, dt=0.08, variant=k))
        # batch 0 = the configuration as BASELINE defines it, batches 1.. = fresh variants of it (other poses, other random streams)
        # step k runs batch k mod nb on capsule k mod S: the two periods must be coprime, or a capsule would meet the same batch
        # again and again (four batches over four -- or two -- capsules: every solve would be a REPEATED batch with exact
        # longest-first history, not a fresh one)
        S_ = max(1, int(n_slots if n_slots is not None else args.streams))
        self.nb = NB_FRESH
        while np.gcd(self.nb, S_) != 1:
            self.nb += 1
        self.host = [gen(k)[:2] for k in range(self.nb + 1)]
        x0, yref = self.host[0]
        assert len(x0) == B
        self.nmom = 16 if cid == 3 else 0

        def make(slot):
            s = solver_factory(N=N, dt=0.08, nsub=3, batch=B, device=local_rank, store_qp_in=(cid == 5))
            s.install_reference_ocp()
            s.set_x0(x0); s.set_yref_all(yref)
            if cid == 3:      # PCE matrix of the 15 Hammersley sigma points (10 terms): K6 runs inside the step
                from tum_control_amd.snmpc import alpha_generation, hammersley_normal, pce_matrix
                s.pce_attach(pce_matrix(hammersley_normal(15, 3), alpha_generation(3, 2)))
            if cid == 5:      # covariance back-off attached to every solve (K7), nominal bounds restored at the start of a step
                from tum_control_amd.r2nmpc import r2_setup
                m, veh = s.cfg["mpc"], s.cfg["veh"]
                S0, BWB = r2_setup(m["stds"], 0.08)
                s.r2_attach(S0, BWB, int(m["uncertainty_propagation_horizon"]), veh["delta_f_min"], veh["delta_f_max"], 1.0)
                s.bounds_snapshot()
            return s

        # the steps are dealt to S capsules in turn, each on its own stream (streaming.py); S = 1: the current torch stream
        S = self.S = max(1, int(n_slots if n_slots is not None else args.streams))
        if self.cuda:
            self.streams = [torch.cuda.current_stream()] if S == 1 else [torch.cuda.Stream() for _ in range(S)]
            self.ring = SolverRing(S, make, [st.cuda_stream for st in self.streams], allow_unstable=True)      # (--streams 5, 6, 8: the measurement of WHY the ring caps at four)
        else:
            self.streams = [None] * S
            self.ring = SolverRing(S, make, allow_unstable=True)
        self.s = self.ring[0]
        self.spp = C["solves_per_step"]
        # result slab gathered to rank 0 each step: (u0[2], cost, status, qp_iter) as 5 doubles per instance, for config 3
        # followed by the PCE mean / variance of x_1 of every scenario group (16 doubles per group): ONE flat buffer, packed
        # on the device by the library, moved with ONE rooted gather. Shards of a strong-scaling job may differ by one group:
        # every rank sends the size of the LARGEST shard (the tail is padding).
        # --gather-iterate: X ((N+1) x 8) and U (N x 2) of every instance follow, in two blocks sized for the LARGEST shard:
        # [summary B x 5 | moments P x nmom | pad][X B x (N+1) x 8 | pad][U B x N x 2 | pad]
        self.with_iterate = bool(getattr(args, "gather_iterate", False))
        self.slab_len = B * 5 + P * self.nmom
        mx = max((sharding.shard_range(groups_total, world, r)[1] - sharding.shard_range(groups_total, world, r)[0]) for r in range(world))
        self.slab_head = mx * gsz * 5 + mx * self.nmom
        self.off_X, self.off_U = self.slab_head, self.slab_head + mx * gsz * (N + 1) * 8
        self.slab_pad = self.slab_head + (mx * gsz * ((N + 1) * 8 + N * 2) if self.with_iterate else 0)
        self.slabs = [torch.zeros(self.slab_pad, dtype=torch.float64, device=dev) for _ in range(S)]      # one per capsule
        self.gather = sharding.ResultGatherer(world, rank, 1, dev, nf=self.slab_pad, ni=1) if self.distributed else None
        # resident copies of the rotated batches
        self.dx0 = [torch.from_numpy(np.ascontiguousarray(h[0])).to(dev) for h in self.host[1:]]
        self.dyr = [torch.from_numpy(np.ascontiguousarray(h[1])).to(dev) for h in self.host[1:]]
        # host-visible legs: every step's results (u0, cost, status, qp_iter -- optionally the whole iterate) land in the capsule's
        # pinned host slab behind an event (tum_ocp_results_async); the host reads them when the capsule's turn comes round again
        self.bind_inputs = bool(getattr(args, "bind_inputs", False))
        self.host_results = None          # None | "summary" | "iterate"
        self.host_ok = self.host_seen = 0
        self.host_checksum = 0.0

    # ---- one step; `marks`: list that receives (solve_begin, solve_end)* and (gather_begin, gather_end) clock marks
    def step(self, marks=None, fresh=None):
        slot, s = self.ring.acquire()
        if self.cuda and self.S > 1:
            with self.torch.cuda.stream(self.streams[slot]):      # (events, the reduction and the gather follow the capsule's stream)
                self._step(slot, s, marks, fresh)
        else:
            self._step(slot, s, marks, fresh)

    def _read(self, res):
        """the host side of a step: read the pinned slab of a finished batch"""
        if res is None:
            return
        summ, X, U = res
        self.host_seen += len(summ); self.host_ok += int((summ[:, 3] == 0).sum())
        self.host_checksum += float(summ[:, 0].sum()) + (float(X[:, 1, 3].sum()) if X is not None else 0.0)

    def drain(self):
        for _, res in self.ring.drain():
            self._read(res)

    def _step(self, slot, s, marks, fresh):
        cid, slab = self.cid, self.slabs[slot]
        if fresh is not None:
            k = fresh % self.nb
            # the step's batch is resident in HBM and copied into the capsule's own arrays first (a copy kernel per field: 8 MB of yref
            # per step on config 2, inside the timed region); --bind-inputs: the capsule reads it where it lies instead
            if self.bind_inputs:
                s.bind_device("x0", self.dx0[k].data_ptr()); s.bind_device("yref", self.dyr[k].data_ptr())
            else:
                s.put_device("x0", self.dx0[k].data_ptr()); s.put_device("yref", self.dyr[k].data_ptr())
        if cid == 5:
            s.bounds_restore()
        s.cold_start()
        for _ in range(self.spp):
            if marks is not None:
                marks.append(self.clock.mark())
            s.solve_async()
            if marks is not None:
                marks.append(self.clock.mark())
        if cid == 3:
            mom_ptr = slab.data_ptr() + 8 * self.B * 5
            s.pce_moments_device("x", 1, mom_ptr, mom_ptr + 8 * self.P * 8)
        if self.host_results:
            self.ring.request_results(slot, self.host_results == "iterate")
            if self.ring.outstanding(slot) == 2:                # the batch this capsule solved S steps ago: read AFTER the new batch and
                self._read(self.ring.take_results(slot))        # its request are on the stream, so the stream never waits for the host
        if self.gather is not None:
            if marks is not None:
                marks.append(self.clock.mark())
            s.get_device("summary", slab.data_ptr())
            if self.with_iterate:
                s.get_device("X", slab.data_ptr() + 8 * self.off_X); s.get_device("U", slab.data_ptr() + 8 * self.off_U)
            self.gather.gather(slab.view(1, -1))
            if marks is not None:
                marks.append(self.clock.mark())

    def barrier(self):
        if self.distributed:
            self.dist.barrier()
        self.clock.sync()

    def timed(self, steps, fresh):
        """EXACTLY `steps` steps between two barriers (+ device sync); returns (max-over-ranks seconds, per-step marks)."""
        self.barrier()
        marks = [[] for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(marks[i], fresh=(i if fresh else None))
        if self.host_results:
            self.drain()                      # (the last results have been READ on the host when the clock stops)
        self.barrier()
        elapsed = time.perf_counter() - t0
        self.rank_elapsed = [elapsed]
        if self.distributed:
            # max over ranks = the job's time; every rank's own time rides along (per-rank rates in the line)
            tt = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.dev)
            every = [self.torch.zeros_like(tt) for _ in range(self.world)]
            self.dist.all_gather(every, tt)
            self.rank_elapsed = [float(t.item()) for t in every]
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, marks

    def solve_ms(self, marks):
        spp = self.spp
        return float(np.mean([self.clock.ms(m[2 * j], m[2 * j + 1]) for m in marks for j in range(spp)]))

    def gather_ms(self, marks):
        if self.gather is None:
            return None
        return float(np.mean([self.clock.ms(m[2 * self.spp], m[2 * self.spp + 1]) for m in marks]))


def run(args, torch, dist, dev, world, rank, local_rank, solver_factory, workload=None, extra_legs=True):
    """The benchmark proper. Returns (result dict, job) on rank 0 (None elsewhere)."""
    job = Job(args, torch, dist, dev, world, rank, local_rank, solver_factory, workload)
    N, cid, B, spp, S = job.N, job.cid, job.B, job.spp, job.S
    for i in range(args.warmup):
        job.step(fresh=None if args.same_batch else i)
    elapsed, marks = job.timed(args.steps, fresh=not args.same_batch)
    rank_elapsed = list(job.rank_elapsed)
    kern_ms = job.solve_ms(marks)             # device time between the events around a solve (S > 1: shared with other batches)
    gat_ms = job.gather_ms(marks)
    # correctness of what was timed: statuses and iteration counts of the batches the capsules solved last
    st = np.concatenate([s.get_stats("status") for s in job.ring]); it = np.concatenate([s.get_stats("qp_iter") for s in job.ring])
    mean_it_fresh = float(it.mean())
    ok_fresh = float((st == 0).mean())

    # ---- the same loop with the results of EVERY step on the host (what a Python caller sees: NMPC_class.py:193-206 reads u0 /
    # cost / status after every solve): results_async behind each solve, the pinned slab read when the capsule comes round
    # again, the last ones inside the timed region. Same capsules, same streams, same fresh batches as `value`.
    hv_value = hv_iter_value = hv_ok = None
    if world == 1 and extra_legs and not args.no_host_legs and hasattr(job.s, "results_async"):
        for mode in ("summary", "iterate"):
            job.host_results = mode; job.host_seen = job.host_ok = 0
            # (first uses are slow and must stay outside the timed region: pinned slabs are allocated when a capsule is first asked
            #  for its results, and the first device-to-host copies of a stream set up its copy path -- several ms each)
            for i in range(int(os.environ.get("BENCH_HOST_WARM", 4 * S))):
                job.step(fresh=i)
            job.drain(); job.host_seen = job.host_ok = 0
            hv_elapsed, hmarks = job.timed(args.steps, fresh=not args.same_batch)
            assert job.host_seen == B * args.steps, (job.host_seen, B, args.steps)
            if os.environ.get("BENCH_DEBUG"):
                print(f"[host leg {mode}] {1e3 * hv_elapsed / args.steps:.3f} ms/step; device ms between the events around each solve: "
                      + " ".join(f"{job.clock.ms(m[0], m[1]):.2f}" for m in hmarks), file=sys.stderr)
            v = B * spp * args.steps / hv_elapsed
            if mode == "summary":
                hv_value, hv_ok = v, job.host_ok / max(1, job.host_seen)
            else:
                hv_iter_value = v
        job.host_results = None

    one_value = one_ms = rep_value = rep_ms = nat_ms = ipm_ms = rep_ipm_ms = None
    mean_it = mean_it_fresh
    U = None
    job1 = job
    legs = world == 1 and extra_legs and not args.no_schedule_legs
    if legs:
        if S > 1:      # the same loop on ONE capsule / ONE stream: a launch has the chip to itself, per-kernel times are clean
            job1 = Job(args, torch, dist, dev, world, rank, local_rank, solver_factory, lambda k: job.host[k], n_slots=1)
            for i in range(args.warmup):
                job1.step(fresh=i)
            one_elapsed, omarks = job1.timed(args.steps, fresh=True)
            one_ms = job1.solve_ms(omarks)
            one_value = B * spp * args.steps / one_elapsed
        else:
            one_ms, one_value = kern_ms, job.global_batch * spp * args.steps / elapsed
        s = job1.s
        # the dominant kernel -- the interior point kernel -- timed on its own (library events around it) over five more steps of
        # the same one-stream fresh-batch loop (what `rocprofv3 --kernel-trace --stats -- python bench.py --streams 1` averages)
        try:
            tm, ti = [], []
            for i in range(5):
                job1.step(fresh=args.steps + i); job1.clock.sync(); tm.append(1e3 * s.get_stats("time_ipm")); ti.append(float(s.get_stats("qp_iter").mean()))
            ipm_ms, mean_it = float(np.median(tm)), float(np.mean(ti))
        except Exception:
            pass
        # (a) the SAME batch every step (batch 0): the longest-first order has this batch's exact iteration counts
        if job1.bind_inputs:          # (back to the capsule's own arrays: the setters below would write through into a resident batch)
            s.bind_device("x0", None); s.bind_device("yref", None)
        s.set_x0(job.host[0][0]); s.set_yref_all(job.host[0][1])
        for _ in range(2):
            job1.step()
        rep_elapsed, rmarks = job1.timed(args.steps, fresh=False)
        rep_ms = job1.solve_ms(rmarks)
        rep_value = B * spp * args.steps / rep_elapsed
        st0 = s.get_stats("status"); it0 = s.get_stats("qp_iter")
        _, U = s.get_iterate()
        rep_ipm_ms = None
        try:
            tm = []
            for _ in range(5):
                job1.step(); job1.clock.sync(); tm.append(1e3 * s.get_stats("time_ipm"))
            rep_ipm_ms = float(np.median(tm))
        except Exception:
            pass
        # (b) the same batch with the instances dispatched in natural order
        s.set_schedule(False)
        nat = [[] for _ in range(5)]
        for m in nat:
            job1.step(m)
        job1.clock.sync()
        nat_ms = float(np.median([job1.clock.ms(m[2 * j], m[2 * j + 1]) for m in nat for j in range(spp)]))
        s.set_schedule(True)

    if rank != 0:
        return None
    C, gsz, P = job.C, job.gsz, job.P
    total = job.global_batch * spp * args.steps
    value = total / elapsed
    shared_yref = gsz > 1
    abytes = algorithmic_bytes(N, warm=(cid == 5), per_instance_yref=not shared_yref)
    # roofline: algorithmic FLOPs of the fresh batches / device time of one solve, from the leg in which a solve has the chip
    # to itself (one stream) when that leg ran; `sustained` = the same FLOPs at the rate of the timed region (`value`)
    roof_ms = one_ms if one_ms is not None else kern_ms
    flops_solve = algorithmic_flops(N, 3, mean_it_fresh)
    ach_tf = flops_solve * B / (roof_ms * 1e-3) / 1e12
    ach_gb = abytes * B / (roof_ms * 1e-3) / 1e9
    sus_tf = flops_solve * value / world / 1e12          # per GPU
    # HBM bytes per launch from the PMC counters cannot be collected from inside this process; they come from the committed
    # rocprofv3 --pmc passes of this same command (profiles/*_traffic*.json), only when the workload matches what was profiled.
    traffic, traffic_src = None, None
    try:
        for tj in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if "_traffic" in f and f.endswith(".json")), reverse=True):
            tr = json.load(open(os.path.join(ROOT, "profiles", tj)))
            if tr.get("batch") == B and tr.get("N") == N and tr.get("config", 2) == cid:
                traffic, traffic_src = tr["traffic_bytes_per_launch"], "profiles/" + tj
                break
    except Exception:
        pass
    dominant = None
    if ipm_ms:
        ipm_flops = mean_it * ipm_flops_per_iter(N) * B
        dominant = {"kernel": "ipm_kernel", "kernel_ms": ipm_ms, "flops_per_launch": ipm_flops, "mean_qp_iter": mean_it,
                    "workload": "fresh batches on one stream (stale longest-first history), five steps behind that leg",
                    "kernel_ms_repeated_batch": rep_ipm_ms,
                    "achieved": ipm_flops / (ipm_ms * 1e-3) / 1e12, "frac": ipm_flops / (ipm_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
        if rep_ipm_ms:           # the same kernel on the repeated batch (exact longest-first order: no tail of late long instances)
            rep_flops = float(it0.mean()) * ipm_flops_per_iter(N) * B
            dominant["frac_repeated_batch"] = rep_flops / (rep_ipm_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
    out = {
        "metric": "SQP-RTI OCP solves/sec (batch), N=40 single-track Pacejka",
        "value": value, "unit": "OCP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{cid - 1}]: {C['name']}, {C['track']} reftraj, cold-start SQP-RTI, one wavefront per OCP; "
                               + ("the same batch every step (--same-batch)" if args.same_batch else
                                  f"{job.nb} differently seeded batches resident in HBM, rotated one per step over {S} capsules (coprime periods: a capsule never meets the batch it solved last)"),
                   "N": N, "nx": 8, "nu": 2, "nsub": 3, "batch_per_gpu": B, "global_batch": job.global_batch,
                   "scenario_group": gsz, "groups_per_gpu": P, "solves_per_step": spp,
                   "streams": S,
                   "collective_backend": (dist.get_backend() if job.distributed else None),
                   "collective_world_size": (dist.get_world_size() if job.distributed else None),
                   "streams_note": "steps are dealt to `streams` capsules in turn, each with its own buffers on its own HIP stream: every "
                                   "step is a complete pass over one fresh batch, consecutive batches overlap on the GPU (`value`); "
                                   "`value_single_stream` is the same loop on one capsule / one stream",
                   "parallelism": f"scenario groups sharded x{world} (a group never straddles ranks), RCCL gather of "
                                  f"{job.slab_pad * 8} B per rank (u0, cost, status, qp_iter"
                                  + (", PCE mean/var of x_1 per group" if job.nmom else "")
                                  + (", the whole iterate X, U" if job.with_iterate else "") + "), one collective per step",
                   "inputs": ("every step's resident batch (x0, yref) is read in place (tum_ocp_bind_device)" if job.bind_inputs else
                              "every step's resident batch (x0, yref) is copied device-to-device into the capsule's arrays inside the timed region (tum_ocp_put_device)"),
                   "gather_iterate": job.with_iterate, "gather_bytes_per_rank": job.slab_pad * 8,
                   "schedule": "workgroups take instances longest-first by the previous solve's IPM iteration count "
                               "(tum_ocp_set_schedule): stale history when every step brings a new batch (`value`, "
                               "`value_single_stream`), exact when the same batch is solved again (`value_repeated_batch`)",
                   "kernel_ms_natural_order": nat_ms,
                   "solves_per_s_per_gpu_natural_order": (B / nat_ms * 1e3) if nat_ms else None,
                   "kernel_ms_repeated_batch": rep_ms},
        "value_single_stream": one_value, "value_repeated_batch": rep_value,
        "value_host_visible": hv_value, "value_host_visible_with_iterate": hv_iter_value, "host_visible_status_ok_frac": hv_ok,
        "host_visible_note": "the loop of `value` with every step's results READ ON THE HOST inside the timed region: u0, cost, status, "
                             "qp_iter of every instance (`value_host_visible`), plus the whole iterate X, U "
                             "(`value_host_visible_with_iterate`), through pinned slabs and an event per capsule (tum_ocp_results_async / "
                             "_wait): the copy of one batch crosses PCIe while the next batches run",
        "solve_ms_per_step": kern_ms * spp, "gather_ms_per_step": gat_ms,
        "per_rank": {"value": [(sharding_size(job, r) * spp * args.steps / t) for r, t in enumerate(rank_elapsed)],
                     "ms_per_step": [1e3 * t / args.steps for t in rank_elapsed],
                     "note": "every rank's own instances / its own wall time between the two barriers (rank 0's gather_ms_per_step is the "
                             "rooted gather as the root sees it); `value` = all instances / the slowest rank's time"},
        "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "pipeline of lin_kernel + cond_kernel + ipm_kernel + expand_kernel (kernel_ms = device time of one solve"
                               + (", one stream: the launches have the chip to themselves)" if (one_ms is not None or S == 1) else
                                  f", {S} streams: launches of consecutive batches share the chip)"),
                     "kernel_ms": roof_ms, "dominant": dominant, "mean_qp_iter": mean_it_fresh,
                     "flops_per_solve": flops_solve,
                     "sustained": {"achieved": sus_tf, "frac": sus_tf / FP64_PEAK_TFLOPS,
                                   "note": "algorithmic FLOPs per solve x `value` per GPU: the rate of the timed region"},
                     "hbm": {"achieved": ach_gb, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_gb / HBM_PEAK_GBPS,
                             "bytes_per_solve": abytes}},
        "status_ok_frac": ok_fresh,
    }
    if rep_value is not None:
        out["status_ok_frac_repeated_batch"] = float((st0 == 0).mean())
    job.U_repeated = U
    return out, job


def sharding_size(job, r):
    """instances of rank r's shard"""
    from tum_control_amd import sharding
    lo, hi = sharding.shard_range(job.groups_total, job.world, r)
    return (hi - lo) * job.gsz


def n38_leg(args, torch, dev, solver_factory):
    """"also report N = 38" (the reference's own horizon, Tp = 3.04 s): fresh-batch rate of config 2 at N = 38."""
    a = argparse.Namespace(**vars(args)); a.horizon = 38; a.config = 2; a.batch = None; a.scaling = "weak"; a.no_schedule_legs = True
    res = run(a, torch, None, dev, 1, 0, dev.index or 0, solver_factory, extra_legs=False)
    out, _ = res
    return {"N": 38, "value": out["value"], "ms_per_step": out["ms_per_step"], "streams": out["config"]["streams"],
            "mean_qp_iter": out["roofline"]["mean_qp_iter"], "status_ok_frac": out["status_ok_frac"]}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` started WITHOUT a launcher (no RANK in the environment): start the N ranks ourselves -- one
    process per GPU, the same script and arguments, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT (a free
    port) in the environment exactly as torch.distributed.run would set them -- and wait for them. Rank 0's ONE JSON line goes to
    our stdout (inherited); a rank that fails takes the others down with it (the exact PIDs started here) and its exit code is
    ours. Under torch.distributed.run (RANK set) this function is never reached: that path is unchanged."""
    import signal
    import subprocess
    n = args.gpus
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # (dmabuf IPC: RCCL between processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdin=subprocess.DEVNULL))
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT", 1800))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                c = p.poll()
                if c is not None:
                    live.remove(p)
                    if c != 0 and rc == 0:
                        rc = c
            if rc != 0 or time.time() > deadline:
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    if rc != 0:
        raise SystemExit(f"bench.py --gpus {n}: a rank exited with code {rc}" if rc != 124 else f"bench.py --gpus {n}: ranks timed out")


def _test_solver_factory():
    """TUM_BENCH_TEST_SOLVER=module:Class (tests only): the control flow of this file on CPU / gloo with a stand-in solver class the
    TEST supplies (tests/test_host_logic.py). The product has no CPU path: without this switch a missing GPU is an error."""
    spec = os.environ.get("TUM_BENCH_TEST_SOLVER")
    if not spec:
        return None
    import importlib
    mod, attr = spec.split(":")
    return getattr(importlib.import_module(mod), attr)


def other_configs_legs(args, torch, dev, solver_factory, steps=8, warmup=3):
    """The other BASELINE configurations and the coupled SNMPC OCP inside the driver-timed line (N = 1 only, a bounded leg each):
    configs 3 / 4 / 5 at their per-GPU size through the SAME `run` as the headline (fresh batches over the default number of capsules, barriers and
    device syncs around `steps` steps), the coupled 88-state SNMPC OCP (N = 38, ten samples) at the shipped propagation
    horizon (uph = 5) and at UPH = Tp (uph = 38): cold start + solve of 4096 instances, one capsule and three in flight."""
    out = {}
    for cid in (3, 4, 5):
        try:
            a = argparse.Namespace(**vars(args)); a.config = cid; a.batch = None; a.scaling = "weak"; a.steps = steps; a.warmup = warmup
            a.no_schedule_legs = True; a.no_host_legs = True
            o, job = run(a, torch, None, dev, 1, 0, dev.index or 0, solver_factory, extra_legs=False)
            out[str(cid)] = {"value": o["value"], "ms_per_step": o["ms_per_step"], "batch": o["config"]["batch_per_gpu"],
                             "solves_per_step": o["config"]["solves_per_step"], "steps": steps, "streams": o["config"]["streams"],
                             "status_ok_frac": o["status_ok_frac"], "mean_qp_iter": o["roofline"]["mean_qp_iter"],
                             "roofline": {"frac": o["roofline"]["sustained"]["frac"], "achieved": o["roofline"]["sustained"]["achieved"],
                                          "unit": "TFLOP/s", "note": "algorithmic FLOPs per solve x value / FP64 peak (rate of the timed region)"},
                             "workload": o["config"]["workload"]}
            del job, o
        except Exception as e:       # (a leg that fails must not take the headline with it; it says so in the line)
            out[str(cid)] = {"error": repr(e)}
    try:
        out.update(snmpc_legs(dev))
    except Exception as e:
        out["snmpc"] = {"error": repr(e)}
    return out


def snmpc_legs(dev, B=4096, N=38, rounds=6):
    from tum_control_amd import config, snmpc as snm
    from tum_control_amd.solver import CoupledSnmpcSolver
    from tum_control_amd.streaming import SolverRing
    from tum_control_amd.workloads import nominal_batch
    import torch
    stds = np.asarray(config.MPC["stds"], dtype=float)
    w = snm.hammersley_normal(10, 3)
    A = snm.pce_matrix(w, snm.alpha_generation(3, 2))
    offs = snm.x0_offsets(w, stds)
    x0, yref = nominal_batch(B, N=N)
    X0 = np.concatenate([x0[:, None, :], x0[:, None, :] + offs[None]], axis=1).reshape(B, -1)
    out = {}
    for uph in (5, 38):
        def mk(_):
            c = CoupledSnmpcSolver(N=N, dt=0.08, batch=B, Apce=A, uph=uph, gamma=config.MPC["gamma"], device=dev.index or 0)
            c.install_reference_ocp()
            c.constraints_set(0, "lbx", X0); c.constraints_set(0, "ubx", X0)
            c.set_yref_all(yref)
            return c
        res = {"N": N, "uph": uph, "n_samples": 10, "batch": B}
        for S in (1, 3):
            ring = SolverRing(S, mk)
            for _ in range(2 * S):
                _, c = ring.acquire(); c.cold_start(); c.solve_async()
            ring.synchronize(); torch.cuda.synchronize()
            K = rounds * S
            t0 = time.perf_counter()
            for _ in range(K):
                _, c = ring.acquire(); c.cold_start(); c.solve_async()
            ring.synchronize(); torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0
            ok = min(float((c.get_stats("status") == 0).mean()) for c in ring)
            it = float(np.mean([c.get_stats("qp_iter").mean() for c in ring]))
            key = "single_capsule" if S == 1 else "three_capsules"
            res[key] = {"value": B * K / dt_, "ms_per_batch": 1e3 * dt_ / K, "batches": K, "status_ok_frac": ok, "mean_qp_iter": it}
            del ring
        res["value"] = res["three_capsules"]["value"]
        out[f"snmpc_uph{uph}"] = res
    return out


def main(argv=None):
    args = parse_args(argv)
    launched = os.environ.get("RANK") is not None             # by torch.distributed.run, or by self_launch below
    if not launched and (args.gpus > 1 or os.environ.get("BENCH_SELF_LAUNCH") == "1"):
        return self_launch(args, argv)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or launched
    stand_in = _test_solver_factory()
    if stand_in is None and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the solver has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: start it as `python bench.py --gpus {args.gpus}` (it launches its "
                         f"own ranks) or as `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}`")
    if stand_in is None:
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible (one rank per GPU)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if stand_in is None:
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")
        assert dist.get_world_size() == args.gpus, f"world size {dist.get_world_size()} != --gpus {args.gpus}"
    if stand_in is None:
        from tum_control_amd.solver import BatchedOcpSolver as factory
    else:
        factory = stand_in

    res = run(args, torch, dist if distributed else None, dev, world, rank, local_rank, factory)
    if rank == 0:
        out, job = res
        out["launch"] = ("self-launched ranks (bench.py --gpus N)" if os.environ.get("BENCH_SELF_LAUNCHED") == "1" else
                         "torch.distributed.run" if launched else "single process")
        if stand_in is not None:
            out["data"] = "stand-in solver on CPU / gloo (control-flow test, not a measurement)"
        gpu_legs = stand_in is None and world == 1
        if gpu_legs and not args.no_schedule_legs and args.horizon != 38 and args.config == 2:
            out["n38"] = n38_leg(args, torch, dev, factory)
        if gpu_legs and not args.no_other_configs and args.config == 2 and args.horizon == 40 and not args.same_batch:
            out["other_configs"] = other_configs_legs(args, torch, dev, factory)
        if not args.no_cpu_baseline and world == 1 and stand_in is None:
            x0, yref = job.host[0]
            cb, u_ref = cpu_baseline(job.N, x0, yref, job.s.cfg)
            out["cpu_baseline"] = cb
            U = job.U_repeated
            if args.config != 5 and U is not None and not args.no_schedule_legs:
                # (the iterate of the repeated-batch leg = batch 0 = what the oracle sample solved; config 5's iterate is
                #  the SECOND solve of the step)
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
