#!/bin/bash
# VGPRs / spills / LDS per kernel of one .hip file (device-only compile; no GPU needed):  tools/kernel_regs.sh conv_kernels.hip [filter]
cd "$(dirname "$0")/../e-osvos_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize $EXTRA --offload-device-only -c "$1" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | awk '
/remark: Function Name:/ {name=$5} / VGPRs: / {v=$4} / AGPRs: / {a=$4} /VGPRs Spill:/ {sp=$5} /ScratchSize/ {p=$5} /LDS Size/ {printf "%s vgpr %s agpr %s spill %s scratch %s lds %s\n", name, v, a, sp, p, $6}' | c++filt | grep -E "${2:-.}"
