#!/usr/bin/env python3
"""Copies the summaries collected by scripts/collect_profiles.sh (gpurun_out/final) into profiles/<tag>_*."""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = "gpurun_out/final", "profiles"
shutil.copy(f"{src}/stats/s_kernel_stats.csv", f"{dst}/{tag}_kernel_stats.csv")          # bench.py --streams 1 (one stream)
if os.path.exists(f"{src}/stats3/s_kernel_stats.csv"):
    shutil.copy(f"{src}/stats3/s_kernel_stats.csv", f"{dst}/{tag}_kernel_stats_3streams.csv")      # the default command
if os.path.exists(f"{src}/stats_rep/s_kernel_stats.csv"):
    shutil.copy(f"{src}/stats_rep/s_kernel_stats.csv", f"{dst}/{tag}_kernel_stats_repeated.csv")      # bench.py --streams 1 --same-batch
for d, n in (("sn38", "snmpc_uph38_kernel_stats.csv"), ("sn05", "snmpc_uph5_kernel_stats.csv"), ("loop26", "loop26_kernel_stats.csv")):
    if os.path.exists(f"{src}/{d}/s_kernel_stats.csv"):
        shutil.copy(f"{src}/{d}/s_kernel_stats.csv", f"{dst}/{tag}_{n}")
for f in ("ab_round3_instruction_work.txt", "fused_expand.txt", "snmpc_prologue_variants.txt", "phase_cycles.txt", "configs.jsonl", "kernel_variants.txt", "ipm_occupancy.txt", "snmpc_bench.txt", "pcie.txt", "streams.txt",
          "closed_loops.txt", "long_horizons.txt", "full_w.txt", "snmpc_sample_counts.txt", "ipm4_phases.txt", "small_batch_variants.txt", "controller_step.txt", "gather_iterate.txt", "ab_saved_builds.txt"):
    if os.path.exists(f"{src}/{f}"):
        shutil.copy(f"{src}/{f}", f"{dst}/{tag}_{f}")
KERNELS = ("lin_kernel", "cond_kernel", "ipm_kernel", "expand_kernel", "nmpc_rti_kernel")
BATCH = {2: 4096, 3: 16384, 4: 16384, 5: 4096}


def traffic(cfg, suffix):
    out = {"kernels": "one solve = lin_kernel + cond_kernel + ipm_kernel + expand_kernel (the pipeline)", "config": cfg, "batch": BATCH[cfg], "N": 40,
           "command": "bench.py --streams 1" + (f" --config {cfg}" if cfg != 2 else ""),
           "units": "KB per launch (rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE)"}
    tot = 0.0
    for d, name in (("pmc_fetch" + suffix, "FETCH_SIZE"), ("pmc_write" + suffix, "WRITE_SIZE")):
        rows = list(csv.DictReader(open(f"{src}/{d}/p_counter_collection.csv")))
        for k in KERNELS:
            v = [float(r["Counter_Value"]) for r in rows if k in r["Kernel_Name"] and r["Counter_Name"] == name]
            if v:
                out[f"{name}_KB_{k}"] = sum(v) / len(v)
                tot += sum(v) / len(v) * (2 if cfg == 5 else 1)      # (config 5: two solves per step, both counted)
    out["traffic_bytes_per_launch"] = tot * 1024 / (2 if cfg == 5 else 1)
    out["traffic_bytes_per_solve"] = out["traffic_bytes_per_launch"] / BATCH[cfg]
    return out


for cfg in (3, 4, 5):
    if os.path.exists(f"{src}/pmc_fetch_c{cfg}/p_counter_collection.csv"):
        json.dump(traffic(cfg, f"_c{cfg}"), open(f"{dst}/{tag}_traffic_c{cfg}.json", "w"), indent=1)
out = {"kernels": "one solve = lin_kernel + cond_kernel + ipm_kernel + expand_kernel (the pipeline; batch > 1024)", "config": 2, "batch": 4096, "N": 40,
       "command": "bench.py --streams 1",
       "units": "KB per launch (rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE)",
       "calibration": "cold_start_kernel in the same runs reads 262 KB (x0) and writes 13 369 344 B (X,U): FETCH_SIZE / WRITE_SIZE "
                      "report these 1:1 for this kernel family's 8-byte-per-lane accesses (the 2x FETCH_SIZE correction of "
                      "MI355X_MICROARCH.md applies to 16-byte-per-lane streams)"}
tot = 0.0
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(f"{src}/{d}/p_counter_collection.csv")))
    for k in KERNELS + ("cold_start",):
        v = [float(r["Counter_Value"]) for r in rows if k in r["Kernel_Name"] and r["Counter_Name"] == name]
        if v:
            out[f"{name}_KB_{k}"] = sum(v) / len(v)
            if k != "cold_start":
                tot += sum(v) / len(v)
out["traffic_bytes_per_launch"] = tot * 1024          # all kernels of one solve
out["traffic_bytes_per_solve"] = out["traffic_bytes_per_launch"] / 4096
for f in os.listdir(dst):
    if f.endswith("_traffic.json") and f.startswith(tag):
        os.remove(os.path.join(dst, f))
json.dump(out, open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
sq = collections.defaultdict(dict)
for d in ("pmc_sq1", "pmc_sq2"):
    rows = list(csv.DictReader(open(f"{src}/{d}/p_counter_collection.csv")))
    acc = collections.defaultdict(list)
    for r in rows:
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, cn), v in acc.items():
        sq[k][cn + "_per_launch"] = sum(v) / len(v)
json.dump(sq, open(f"{dst}/{tag}_sq_counters.json", "w"), indent=1)
b = json.load(open(f"{src}/bench.json"))
b["roofline"]["traffic"] = out["traffic_bytes_per_launch"]; b["roofline"]["traffic_source"] = f"profiles/{tag}_traffic.json"
json.dump(b, open(f"{dst}/{tag}_bench.json", "w"))
print(tag, "value", b["value"], "one stream", b.get("value_single_stream"), "repeated", b.get("value_repeated_batch"), "ms", b["ms_per_step"], "frac", b["roofline"]["frac"],
      "traffic/solve", out["traffic_bytes_per_solve"], "cpu", b["cpu_baseline"]["value"])
