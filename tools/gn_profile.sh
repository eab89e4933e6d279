#!/bin/bash
# tools/gn_profile.sh OUTDIR [BATCH]: kernel stats of a few GroupNorm-mode fine-tune steps (rocprofv3 --kernel-trace --stats)
O=$PWD/gpurun_out/$1; B=${2:-3}; mkdir -p $O
export TMPDIR=/tmp
EOSVOS_STEP_NORM=gn rocprofv3 --kernel-trace --stats --output-format csv -d $O/gn$B -- python3 tools/step_profile.py $B > /dev/null 2> $O/gn_prof.err
find $O/gn$B -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/gn_b${B}_kernel_stats.csv
rm -rf $O/gn$B
head -40 $O/gn_b${B}_kernel_stats.csv
