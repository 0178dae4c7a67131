#!/bin/bash
# kernel trace of one band + the per-stage arithmetic: bash tools_band_trace2.sh <tag> <workload> <N>   (environment passes through)
TAG=$1; WL=$2; N=$3
OUT=/root/repo/gpurun_out/bandtrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 /root/repo/tools/tools_band_time.py --workload $WL --splits $N --steps 10 > $OUT/out.txt 2> $OUT/err.txt
python3 /root/repo/tools/tools_band_stage_times.py $(ls $OUT/*/*_kernel_trace.csv | head -1)
