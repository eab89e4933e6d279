#!/bin/bash
# interleaved step-time A/B of environment switches on ONE box:  tools/ab_env.sh "" "EOSVOS_NO_WGRAD_GROUP=1" "EOSVOS_LIB=..."   (3 rounds)
for r in 1 2 3; do
  for v in "$@"; do
    echo "round $r [${v:-default}]: $(env $v python tools/steptime.py 2>&1 | grep ms/step | tr '\n' ' ')"
  done
done
