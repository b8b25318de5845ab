#!/usr/bin/env python3
"""Copies the summaries collected by scripts/collect_profiles.sh (gpurun_out/final) into profiles/<tag>_*."""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
src, dst = "gpurun_out/final", "profiles"
shutil.copy(f"{src}/stats/s_kernel_stats.csv", f"{dst}/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/phase_cycles.txt", f"{dst}/{tag}_phase_cycles.txt")
shutil.copy(f"{src}/configs.txt", f"{dst}/{tag}_configs.txt")
if os.path.exists(f"{src}/snmpc_bench.txt"):
    shutil.copy(f"{src}/snmpc_bench.txt", f"{dst}/{tag}_snmpc_bench.txt")
    shutil.copy(f"{src}/sn_stats/s_kernel_stats.csv", f"{dst}/{tag}_snmpc_kernel_stats.csv")
if os.path.exists(f"{src}/schedules.json"):
    shutil.copy(f"{src}/schedules.json", f"{dst}/{tag}_schedules.json")
out = {"kernel": "nmpc_rti_kernel<false>", "batch": 4096, "N": 40,
       "units": "KB per launch (rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE)",
       "calibration": "cold_start_kernel in the same runs reads 262 KB (x0) and writes 13 369 344 B (X,U): FETCH_SIZE / WRITE_SIZE "
                      "report these 1:1 for this kernel family's 8-byte-per-lane accesses (the 2x FETCH_SIZE correction of "
                      "MI355X_MICROARCH.md applies to 16-byte-per-lane streams)"}
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(f"{src}/{d}/p_counter_collection.csv")))
    v = [float(r["Counter_Value"]) for r in rows if "nmpc_rti" in r["Kernel_Name"] and r["Counter_Name"] == name]
    c = [float(r["Counter_Value"]) for r in rows if "cold_start" in r["Kernel_Name"] and r["Counter_Name"] == name]
    out[name + "_KB"] = sum(v) / len(v); out[name + "_n"] = len(v); out[name + "_cold_start_KB"] = sum(c) / len(c)
out["traffic_bytes_per_launch"] = (out["FETCH_SIZE_KB"] + out["WRITE_SIZE_KB"]) * 1024
out["traffic_bytes_per_solve"] = out["traffic_bytes_per_launch"] / 4096
for f in os.listdir(dst):
    if f.endswith("_traffic.json"):
        os.remove(os.path.join(dst, f))
json.dump(out, open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
sq = {}
for d in ("pmc_sq1", "pmc_sq2"):
    rows = list(csv.DictReader(open(f"{src}/{d}/p_counter_collection.csv")))
    acc = collections.defaultdict(list)
    for r in rows:
        if "nmpc_rti" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        sq[k + "_per_wave"] = sum(v) / len(v) / 4096
json.dump(sq, open(f"{dst}/{tag}_sq_counters.json", "w"), indent=1)
b = json.load(open(f"{src}/bench.json"))
b["roofline"]["traffic"] = out["traffic_bytes_per_launch"]; b["roofline"]["traffic_source"] = f"profiles/{tag}_traffic.json"
json.dump(b, open(f"{dst}/{tag}_bench.json", "w"))
print(tag, "value", b["value"], "ms", b["ms_per_step"], "frac", b["roofline"]["frac"], "traffic/solve", out["traffic_bytes_per_solve"], "cpu", b["cpu_baseline"]["value"])
print(sq)
