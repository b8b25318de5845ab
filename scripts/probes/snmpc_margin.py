#!/usr/bin/env python3
"""Largest deviations GPU vs oracle over the configurations of tests/test_snmpc.py (how much margin the 1e-7 tolerance has)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401  (package alias)
import test_snmpc as T
from oracle import oracle as orc

gd = os.path.join(ROOT, "tests", "golden")
orig = np.testing.assert_allclose
worst = {}
def spy(a, b, rtol=1e-7, atol=0, err_msg="", **kw):
    a = np.asarray(a, float); b = np.asarray(b, float)
    r = np.abs(a - b) / (atol + rtol * np.abs(b)) if (atol + rtol) > 0 else np.abs(a - b)
    key = err_msg.split(" solve")[0] if err_msg else "other"
    worst[key] = max(worst.get(key, 0.0), float(np.max(r)))
np.testing.assert_allclose = spy
for N, uph in [(38, 5), (38, 15), (40, 5), (38, 0), (40, 1), (12, 12)]:
    worst.clear()
    T._gpu_vs_oracle(gd, N, uph, poses=[0, 26, 30])
    print(N, uph, "max |diff| / tolerance:", {k: round(v, 4) for k, v in worst.items()})
