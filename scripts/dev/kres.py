#!/usr/bin/env python3
"""Kernel resource table of a build: name, VGPRs, AGPRs, spills, scratch, LDS, occupancy (hipcc -Rpass-analysis=kernel-resource-usage).
usage: kres.py [-o out.so] [-D...] [filter]   -- compiles tum-control_amd/csrc/tum_nmpc.hip for gfx950"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "tum-control_amd", "csrc")
out, defs, flt, src = "/tmp/kres.so", [], None, "tum_nmpc.hip"
a = sys.argv[1:]
while a:
    x = a.pop(0)
    if x == "-o": out = a.pop(0)
    elif x == "-s": src = a.pop(0)
    elif x.startswith("-D") or x.startswith("-mllvm") or x.startswith("-O"): defs.append(x)
    else: flt = x
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
       "-Rpass-analysis=kernel-resource-usage", "-o", out, src] + defs
r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur: rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
def dem(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
print("%-46s %5s %5s %6s %6s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "sgprSp", "vgprSp", "scratch", "LDS", "occ"))
for c in rows:
    n = dem(c["name"]).replace("tum::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    if flt and flt not in n: continue
    print("%-46s %5s %5s %6s %6s %7s %6s %4s" % (n[:46], c.get("VGPRs", "?"), c.get("AGPRs", "?"), c.get("SGPRs Spill", c.get("Spilled SGPRs", "?")),
          c.get("VGPRs Spill", c.get("Spilled VGPRs", "?")), c.get("ScratchSize [bytes/lane]", "?"), c.get("LDS Size [bytes/block]", "?"), c.get("Occupancy [waves/SIMD]", "?")))
