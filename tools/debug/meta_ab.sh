#!/bin/bash
# interleaved A/B of environment switches on the meta metric (tasks/s at 1 and 4 tasks per rank): tools/debug/meta_ab.sh "" "EOSVOS_X=1"
for r in 1 2 3; do
  for v in "$@"; do
    for t in 1 4; do
      x=$(env $v python bench.py --metric meta --tasks-per-rank $t --steps 30 --warmup 3 --no-cpu-baseline --no-ab 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
      echo "round $r [${v:-default}] tasks per rank $t: $x tasks/s"
    done
  done
done
