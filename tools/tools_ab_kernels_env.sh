#!/bin/bash
# per-kernel durations (single stream) of bench.py --workload c4 under rocprofv3, one build, several environments on one box:
# bash tools/tools_ab_kernels_env.sh base VAR=1 ...
for v in "$@"; do
  rm -rf /root/repo/gpurun_out/ab_k; mkdir -p /root/repo/gpurun_out/ab_k; cd /tmp; export TMPDIR=/tmp
  if [ "$v" = base ]; then E="X_UNUSED=1"; else E="$v"; fi
  env $E GCM_PE_SINGLE_STREAM=1 GCM_BENCH_MIN_TIMED_S=0.1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ab_k -- python3 /root/repo/bench.py --no-cpu --only --workload ${WL:-c4} --steps 8 --warmup 2 > /dev/null 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob
f=glob.glob("/root/repo/gpurun_out/ab_k/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "gcm::pe_" in r["Name"] and "to_" not in r["Name"]: print("%-60s %8.1f us x %s" % (r["Name"][:60], float(r["AverageNs"])/1e3, r["Calls"]))
PY
done
