#!/bin/bash
# A/B of two builds of libgcmcore.so on one box, alternated: bash tools_ab.sh <alt .so> <reps> [bench args...]
ALT=$1; REPS=$2; shift 2
for i in $(seq $REPS); do
  for lib in default $ALT; do
    if [ $lib = default ]; then unset GCMCORE_LIB; else export GCMCORE_LIB=$lib; fi
    python3 /root/repo/bench.py --no-cpu --only "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],4), round(d['ms_per_step_min'],4))"
  done
done
