#!/bin/bash
# N = 8 band of c4 under the orchestration switches and an emulated exchange time: bash tools/tools_band_matrix.sh <reps> [delays...]
REPS=$1; shift; DELAYS=${@:-0}
for rep in $(seq $REPS); do
 for d in $DELAYS; do
  for split in 1 0; do
   for stop in 1 0; do
    ms=$(GCM_PE_K1_SPLIT=$split GCM_PE_STOP_EVENTS=$stop GCM_BAND_EXCHANGE_DELAY_US=$d python3 /root/repo/tools/tools_band_time.py --workload c4 --splits 8 --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['bands'][0]['ms_per_step'],4))")
    echo "delay_us=$d k1_split=$split stop_events=$stop  N=8 band ms/step $ms"
   done
  done
 done
done
