"""dev: the SNMPC closed loop on the device at a small batch, plain launches (below the graph threshold) for the kernel trace"""
import sys
sys.path.insert(0, '/root/repo')
import torch  # noqa: F401
from tum_control_amd.closed_loop import ClosedLoopBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cl = ClosedLoopBatch("monteblanco", batch=B, N=38, Tp=3.04, controller="snmpc", on_device=True, log_capacity=0)
cl.dev.run(40)
