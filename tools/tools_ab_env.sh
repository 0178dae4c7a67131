#!/bin/bash
# A/B of one build under two environments on one box, alternated:
#   bash tools_ab_env.sh <VAR=value> <reps> <command...>     (A: the variable unset, B: set)
# prints the last line of the command's stdout for each run
SET=$1; REPS=$2; shift 2
for i in $(seq $REPS); do
  echo "A(default): $("$@" 2>/dev/null | tail -1)"
  echo "B($SET): $(env $SET "$@" 2>/dev/null | tail -1)"
done
