#!/usr/bin/env python3
"""Host-CPU diagnostics for the cpu_baseline leg: cores visible, cgroup quota, oracle thread scaling."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import OracleOcp
from tum_control_amd.workloads import nominal_batch
from tum_control_amd import config
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
os.system("grep -m1 'model name' /proc/cpuinfo; nproc")
N = 40
x0, yref = nominal_batch(2048, N=N)
m = config.MPC
o = OracleOcp(N, 0.08, 3)
o.set_weights(m["q_lon"], m["q_yaw"], m["q_vel"], m["r_jerk"], m["r_steering_rate"], m["L1_pen"], m["L2_pen"], scale=0.01)
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1): break
    n = min(2048, 128 * nt)
    t = time.perf_counter(); o.solve_batch_cold(x0[:n], yref[:n], nt); dt = time.perf_counter() - t
    print(f"threads {nt:4d}: {n/dt:9.1f} solves/s  ({n/dt/nt:7.1f} per thread)")
