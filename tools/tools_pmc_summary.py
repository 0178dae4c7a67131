#!/usr/bin/env python3
"""Summarise a tools_prof.sh output directory: per-kernel stats + PMC averages."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/trace/*/*_kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    print(open(f).read().strip())
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
    files = sorted(glob.glob("%s/%s/*/*_counter_collection.csv" % (d, sub)), key=__import__("os").path.getmtime)
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if kern in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("== %s (per-dispatch average, n)" % sub)
    for (k, c), v in sorted(acc.items()):
        print("%-62s %-22s %16.1f  n=%d" % (k, c, sum(v) / len(v), len(v)))
