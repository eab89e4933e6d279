#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=1 ..."  ->  e-osvos_amd/variants/libeosvos_NAME.so (A/B builds for tools/ab_libs.sh)
set -e
N=$1; F=$2; D=$(mktemp -d); S=$PWD/e-osvos_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $F -I$S"
/opt/rocm/bin/hipcc $FL -c $S/conv_kernels.hip -o $D/conv.o &
/opt/rocm/bin/hipcc $FL -c $S/misc_kernels.hip -o $D/misc.o &
/opt/rocm/bin/hipcc $FL -x hip -c $S/engine.cpp -o $D/engine.o &
wait
mkdir -p e-osvos_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o e-osvos_amd/variants/libeosvos_$N.so $D/conv.o $D/misc.o $D/engine.o
rm -rf $D; ls -la e-osvos_amd/variants/libeosvos_$N.so
