#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE PMC passes of tools_prof.sh into profiles/<round>/traffic.json:
HBM bytes per launch of each kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts 64 B per 128-B request: x2; both counters are in KiB... reported here in
bytes = KB * 1024).  The x2 was calibrated on this build's own pe_to_device_kernel, a pure
199 MB streaming read with 8-B-per-lane loads: FETCH_SIZE read 97.2 MB for it."""
import collections
import csv
import glob
import json
import os
import sys

prof_dir, out_path, workload = sys.argv[1], sys.argv[2], sys.argv[3]
res = collections.defaultdict(dict)
for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = sorted(glob.glob("%s/%s/*/*_counter_collection.csv" % (prof_dir, sub)), key=os.path.getmtime)
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if r["Counter_Name"] == key and "gcm::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][key + "_KB_per_launch"] = sum(v) / len(v)
for k, d in res.items():
    f, w = d.get("FETCH_SIZE_KB_per_launch"), d.get("WRITE_SIZE_KB_per_launch")
    if f is not None and w is not None:
        d["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        d["correction"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)"
data = {}
if os.path.exists(out_path):
    data = json.load(open(out_path))
data[workload] = res
json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(data[workload], indent=1))
