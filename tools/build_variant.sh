#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..."  ->  e-osvos_amd/variants/libeosvos_NAME.so  (A/B tuning builds)
set -e
cd "$(dirname "$0")/../e-osvos_amd/csrc"
mkdir -p ../variants /tmp/var_$1
for f in conv_kernels.hip misc_kernels.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -c $f -o /tmp/var_$1/${f%.hip}.o &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -x hip -c engine.cpp -o /tmp/var_$1/engine.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libeosvos_$1.so /tmp/var_$1/*.o
echo built ../variants/libeosvos_$1.so
