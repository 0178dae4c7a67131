#!/bin/bash
# kernel timeline of one step in the middle of a bench run: bash tools_step_trace.sh <tag> [bench args...]
TAG=$1; shift
OUT=/root/repo/gpurun_out/steptrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /root/repo/bench.py --no-cpu --only "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [i for i, r in enumerate(rows) if "spu_filter" in r["Kernel_Name"] or "fused" in r["Kernel_Name"]]
i0 = k[len(k) // 2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 24]:
    print("%-40s q=%s start %8.1f end %8.1f dur %7.1f us grid %s" % (r["Kernel_Name"].replace("void gcm::", "")[:40], r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size"))))
PY
