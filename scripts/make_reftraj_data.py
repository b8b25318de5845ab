#!/usr/bin/env python3
"""Packs the race-line reference trajectories the BASELINE configs name (Monteblanco, LVMS, Modena;
/root/reference/Trajectories/reftraj_*_edgar.json: keys pos_x,pos_y,ref_v,ref_yaw) into one small
binary data file for bench.py / tests. Data only; runs in the build container."""
import json, os
import numpy as np
REF = "/root/reference/Trajectories"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tum-control_amd", "data", "reftraj.npz")
out = {}
for t in ("monteblanco", "lvms", "modena"):
    with open(os.path.join(REF, f"reftraj_{t}_edgar.json")) as f:
        d = json.load(f)
    out[t] = np.stack([np.asarray(d[k], float) for k in ("pos_x", "pos_y", "ref_yaw", "ref_v")], axis=1)
np.savez_compressed(OUT, **out)
print({k: v.shape for k, v in out.items()}, os.path.getsize(OUT))
