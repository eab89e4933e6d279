#!/bin/bash
# tools/variant_ab.sh "v1 v2 ..." [layers]: hot kernel, chosen layers and step time for the default build and variants
L=${2:-60,61,54,44,46,25,12,3}
for v in default $1; do
  if [ $v = default ]; then unset EOSVOS_LIB; else export EOSVOS_LIB=$PWD/e-osvos_amd/variants/libeosvos_$v.so; fi
  echo "== $v"
  python tools/kernel_ab.py 2 3 $L 2>&1 | tail -9
  python tools/steptime.py
done
