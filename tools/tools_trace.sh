#!/bin/bash
# kernel trace + stats only: bash tools_trace.sh <tag> [bench args...]
TAG=$1; shift
OUT=/root/repo/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /root/repo/bench.py --no-cpu "$@" > $OUT/bench.json 2> $OUT/err.txt
cat $OUT/*/*_kernel_stats.csv
