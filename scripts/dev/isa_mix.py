#!/usr/bin/env python3
"""Static instruction mix of a kernel of the library, phase by phase.
  isa_mix.py <kernel name substring> [extra hipcc flags ...]
compiles csrc/tum_nmpc.hip to gfx950 assembly (device only), takes the kernel whose mangled name contains the substring and counts
instruction classes between consecutive s_memtime instructions (the instrumented interior point kernel ipm_kernel<true, 5>: its phase
timers; a kernel without timers is one segment). Used for profiles/r05_ipm_instruction_mix.txt."""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sub, extra = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only",
                    *extra, "-o", out, "tum_nmpc.hip"], cwd=os.path.join(ROOT, "tum-control_amd", "csrc"), check=True, capture_output=True)
    t = open(out).read()
names = [n for n in re.findall(r"^(_Z\w+):", t, re.M) if sub in n]
assert names, "no kernel matches"
KEYS = ["mfma16", "mfma4", "fp64", "cndmask", "dpp", "accvgpr", "mov", "xlane", "v_int", "lds", "vmem", "s_nop", "waitcnt", "salu"]
def cls(op):
    if op.startswith("v_mfma_f64_16"): return "mfma16"
    if op.startswith("v_mfma"): return "mfma4"
    if op.startswith("v_") and "f64" in op: return "fp64"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_mov_b32_dpp"): return "dpp"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("v_mov"): return "mov"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")): return "xlane"
    if op.startswith("v_"): return "v_int"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_")): return "vmem"
    if op == "s_nop": return "s_nop"
    if op.startswith("s_waitcnt"): return "waitcnt"
    return "salu"
for name in names:
    i = t.index(name + ":"); j = t.index(".Lfunc_end", i)
    body = [x.strip() for x in t[i:j].split("\n")]
    body = [x for x in body if x and not x.startswith((".", ";"))]
    segs, cur = [], Counter()
    for x in body:
        op = x.split()[0]
        if op.startswith("s_memtime"):
            segs.append(cur); cur = Counter(); continue
        cur[cls(op)] += 1
    segs.append(cur)
    print(name, f"({len(body)} instructions)")
    print("seg  total " + " ".join(f"{k:>7s}" for k in KEYS))
    for n, c in enumerate(segs):
        tot = sum(c.values())
        if tot >= 20:
            print(f"{n:3d} {tot:6d} " + " ".join(f"{c[k]:7d}" for k in KEYS))
