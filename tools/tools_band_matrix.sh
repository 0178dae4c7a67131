#!/bin/bash
# N = 8 band of c4 under the orchestration switches and an emulated exchange time:
#   bash tools/tools_band_matrix.sh <reps> "<VAR=0 settings to try, one per word; 'base' = none>" [delays...]
REPS=$1; VARIANTS=$2; shift 2; DELAYS=${@:-0}
for rep in $(seq $REPS); do
 for d in $DELAYS; do
  for v in $VARIANTS; do
    if [ "$v" = base ]; then E=""; else E="$v"; fi
    ms=$(env $E GCM_BAND_EXCHANGE_DELAY_US=$d python3 /root/repo/tools/tools_band_time.py --workload c4 --splits 8 --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['bands'][0]['ms_per_step'],4))")
    echo "exchange_delay_us=$d  $v  N=8 band ms/step $ms"
  done
 done
done
