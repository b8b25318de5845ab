#!/usr/bin/env python3
"""Development aid: closed-loop weight sweep vs logged acados, error growth per step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tum_control_amd.closed_loop import ClosedLoopBatch
d = np.load(os.path.join(ROOT, "tests/golden/closed_loop_monteblanco_150.npz"))
cl = ClosedLoopBatch("monteblanco", batch=26, params=d["params"], N=38, Tp=3.04)
log = cl.run(150)
C = d["CiLX"].copy(); C[:, :, 2] = np.unwrap(C[:, :, 2], axis=1)
U = log["simU"].transpose(1, 0, 2); X = log["CiLX"].transpose(1, 0, 2)
eu = np.abs(U - d["simU"]).max(axis=2); ex = np.abs(X - C).max(axis=2)
print("status nonzero:", (log["simSolverDebug"][:, :, 4] != 0).sum())
for i in (0, 1, 2, 3, 5, 10, 20, 50, 100, 149):
    print(i, "eu max %.2e (inst %d)" % (eu[:, i].max(), eu[:, i].argmax()), "ex max %.2e" % ex[:, i + 1].max())
print("worst instances by eu:", np.argsort(-eu.max(axis=1))[:5], np.sort(eu.max(axis=1))[::-1][:5])
