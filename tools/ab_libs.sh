#!/bin/bash
# interleaved step-time A/B of library builds on ONE box:  tools/ab_libs.sh default trunc ...   (3 rounds)
for r in 1 2 3; do
  for v in "$@"; do
    if [ $v = default ]; then unset EOSVOS_LIB; else export EOSVOS_LIB=$PWD/e-osvos_amd/variants/libeosvos_$v.so; fi
    echo "round $r $v: $(python tools/steptime.py 2>&1 | grep ms/step | tr '\n' ' ')"
  done
done
