#!/bin/bash
# Produces the per-round evidence under gpurun_out/: bench line, rocprofv3 kernel stats of the same
# command, per-layer report.  usage: tools/round_artifacts.sh r01
R=${1:-r01}
O=$PWD/gpurun_out/$R; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py > $O/prof_bench.json 2> $O/prof.err
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -12 $O/kernel_stats.csv
EOSVOS_NO_SIDE_STREAM=1 EOSVOS_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d $O/lt -- python3 tools/step_profile.py 3 2> $O/trace3.log > /dev/null
python3 tools/layer_report.py $(find $O/lt -name "*kernel_trace.csv" | head -1) $O/trace3.log > $O/layer_report_b3.txt 2>&1
tail -5 $O/layer_report_b3.txt
rm -rf $O/lt $O/prof
