#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (separate runs).
# usage: bash tools_prof.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --no-cpu "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 /root/repo/bench.py --no-cpu "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 /root/repo/bench.py --no-cpu "$@" > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 /root/repo/bench.py --no-cpu "$@" > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- python3 /root/repo/bench.py --no-cpu "$@" > /dev/null 2> $OUT/pmc_tcc.err
python3 /root/repo/tools/tools_pmc_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt | head -40
