#!/usr/bin/env python3
"""kstat.py <rocprofv3 output dir> <substring>: calls and average duration (us) of the kernels whose name contains the substring (s_kernel_stats.csv)"""
import csv, glob, sys
d, sub = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]:
            print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.2f} us")
