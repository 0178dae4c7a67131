#!/usr/bin/env python3
"""Per-stage arithmetic of a band's kernel trace (rocprofv3 --kernel-trace csv of tools_band_time.py): for the
interior rows' update kernel (the long launch of every Euler stage) the period between two launches, split into
'previous K4 end -> K2a start', K2a, K3, 'K3 end -> K4 start', K4; medians over the steady steps of the run.
  python3 tools_band_stage_times.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3, int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)) for r in rows]
upd = [e for e in ev if "pe_update_rows" in e[0]]
big = max(e[3] for e in upd)
k4 = [e for e in upd if e[3] == big]
k4 = k4[len(k4) // 4: -2]                      # steady part
stages = {"true": [], "false": []}             # SAME = true: predictor
for a, b in zip(k4[:-1], k4[1:]):
    inside = [e for e in ev if a[2] - 1 <= e[1] and e[2] <= b[2] + 1]
    k2a = [e for e in inside if "pe_geopot_kernel" in e[0] and e[3] > 20000]
    k3 = [e for e in inside if "pe_pgf_filter" in e[0]]
    if len(k2a) != 1 or len(k3) != 1:
        continue
    kind = b[0].split("pe_update_rows_kernel<")[1].split(">")[0].split(",")[2].strip()      # <T, R, SAME, ...>: SAME = predictor
    stages[kind].append((k2a[0][1] - a[2], k2a[0][2] - k2a[0][1], k3[0][1] - k2a[0][2], k3[0][2] - k3[0][1], b[1] - k3[0][2], b[2] - b[1], b[2] - a[2]))
names = ("K4end->K2a", "K2a", "K2a->K3", "K3", "K3end->K4", "K4 interior", "stage")
tot = 0.0
for kind, label in (("true", "predictor"), ("false", "corrector")):
    s = stages[kind]
    if not s:
        continue
    m = [sorted(x[i] for x in s)[len(s) // 2] for i in range(len(names))]       # medians (a run boundary is a host gap)
    tot += m[-1]
    print("%-9s (%3d stages): " % (label, len(s)) + "  ".join("%s %.1f" % (n, v) for n, v in zip(names, m)))
print("step = %.1f us (under the profiler)" % tot)
