import sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from tum_control_amd.closed_loop import ClosedLoopBatch
cl = ClosedLoopBatch("monteblanco", batch=int(sys.argv[1]) if len(sys.argv) > 1 else 4096, N=38, Tp=3.04, on_device=True, log_capacity=0)
cl.dev.run(40)   # below the graph threshold: plain launches, visible to the kernel trace
