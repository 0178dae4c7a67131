#!/bin/bash
# one gpurun call: rocprofv3 summaries + PMC traffic of the bench workloads -> gpurun_out/r03/prof_*
# usage: bash tools/tools_collect_profiles.sh <round dir under gpurun_out> <workload>...
R=$1; shift
mkdir -p /root/repo/gpurun_out/$R
for w in "$@"; do
  case $w in
    c3) ARGS="--steps 20 --warmup 5 --only";;
    c2) ARGS="--workload c2 --steps 600 --warmup 50 --only";;
    c4) ARGS="--workload c4 --steps 16 --warmup 3 --only";;
    c4_f32) ARGS="--workload c4_f32 --steps 16 --warmup 3 --only";;
    c5_phys) ARGS="--workload c5_phys --steps 6 --warmup 2 --only";;
  esac
  bash /root/repo/tools/tools_prof.sh $w $ARGS > /root/repo/gpurun_out/$R/${w}_rocprofv3_summary.txt 2>&1
  cp /root/repo/gpurun_out/prof_$w/bench_trace.json /root/repo/gpurun_out/$R/${w}_bench_under_rocprofv3.json
  python3 /root/repo/tools/tools_traffic.py /root/repo/gpurun_out/prof_$w /root/repo/gpurun_out/$R/traffic.json $w > /dev/null
  echo "done $w"
done
