#!/bin/bash
# Produces the per-round evidence under gpurun_out/$R: bench lines (fine-tune + meta), rocprofv3 kernel stats of the
# same bench command, per-layer reports, PMC passes of the dominant kernel.  usage: tools/round_artifacts.sh r02
R=${1:-r02}
O=$PWD/gpurun_out/$R; mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/bench_b3.json 2> $O/bench.err; tail -c 600 $O/bench_b3.json
python3 bench.py --metric meta > $O/bench_meta_b1.json 2> $O/bench_meta.err; tail -c 300 $O/bench_meta_b1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-meta --no-ab > $O/prof_bench.json 2> $O/prof.err
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/bench_b3_kernel_stats.csv
head -8 $O/bench_b3_kernel_stats.csv
rm -rf $O/prof
tools/layer_prof.sh $R 3 > /dev/null 2>&1
tools/layer_prof.sh $R 1 > /dev/null 2>&1
K=$(python3 -c "import json;print(json.load(open('$O/bench_b3.json'))['roofline']['kernel'])")
tools/pmc_dominant.sh $R "$K" 3
