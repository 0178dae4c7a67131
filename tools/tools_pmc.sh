#!/bin/bash
# one PMC pass: bash tools_pmc.sh <tag> "<counters>" [bench args...]
TAG=$1; CTR=$2; shift; shift
OUT=/root/repo/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --output-format csv -d $OUT -- python3 /root/repo/bench.py --no-cpu --only "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "rocclr" in k or "diag" in k: continue
    print("%-52s %-26s %16.1f n=%d" % (k, c, sum(v)/len(v), len(v)))
PY
