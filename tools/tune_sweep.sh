#!/bin/bash
# step time at batch 1 / 3 for a list of env settings:  tools/tune_sweep.sh "A=1 B=2" "A=3" ...
for cfg in "" "$@"; do
  echo "== ${cfg:-default}"
  env $cfg python tools/steptime.py 2>&1 | grep "ms/step"
done
