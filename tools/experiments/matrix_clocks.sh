#!/bin/bash
# the distance matrix timed in several fresh processes, with the clocks rocm-smi reports while each one runs
R=$GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  ( TPBS="0" bash $R/tools/experiments/matrix_tpb.sh 2>&1 | tail -1 ) &
  sleep 2.2
  /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | head -12 | tr "\n" ";"
  echo
  wait
done
