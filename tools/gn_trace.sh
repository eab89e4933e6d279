#!/bin/bash
# tools/gn_trace.sh OUTDIR [BATCH]: per-launch durations of the GroupNorm passes of one iteration, grouped by grid size
O=$PWD/gpurun_out/$1; B=${2:-3}; mkdir -p $O
export TMPDIR=/tmp
EOSVOS_STEP_NORM=gn rocprofv3 --kernel-trace --output-format csv -d $O/gt$B -- python3 tools/step_profile.py $B > /dev/null 2> $O/gn_trace.err
python3 - "$(find $O/gt$B -name '*kernel_trace.csv' | head -1)" > $O/gn_trace_b$B.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'gn_' not in n:
        continue
    key = (n.split('(')[0].replace('void eosvos::', ''), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(agg):
    v = agg[k]
    print('%-28s grid %4d x %d x %d  launches %4d  avg %.1f us  min %.1f' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v)))
PY
rm -rf $O/gt$B
cat $O/gn_trace_b$B.txt
