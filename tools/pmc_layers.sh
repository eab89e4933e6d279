#!/bin/bash
# Per-launch SQ counters of every matrix kernel of a single-stream step, joined with the engine's launch log:
#   tools/pmc_layers.sh OUTDIR [batch]   ->  gpurun_out/OUTDIR/pmc_layers_b$B.txt
# (one rocprofv3 --pmc pass with --kernel-trace only: gpurun-safe)
O=$PWD/gpurun_out/$1; B=${2:-3}; mkdir -p $O
export TMPDIR=/tmp
EOSVOS_NO_SIDE_STREAM=1 EOSVOS_TRACE=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $O/pl$B -- python3 tools/step_profile.py $B 2> $O/pl_trace$B.log > /dev/null
python3 - "$O" "$B" <<'PY' > $O/pmc_layers_b$B.txt 2>&1
import csv, glob, re, sys, collections
O, B = sys.argv[1], sys.argv[2]
cc = glob.glob(f'{O}/pl{B}/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob(f'{O}/pl{B}/**/*kernel_trace.csv', recursive=True)[0]
disp = {}
for r in csv.DictReader(open(kt)):
    disp[r['Dispatch_Id']] = r
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    cnt[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
rows = sorted(disp.values(), key=lambda r: int(r['Start_Timestamp']))
ismf = lambda n: 'conv_x6_kernel' in n or 'wgrad_x6_kernel<' in n or 'wgrad_x6_group_kernel<' in n
mf = [r for r in rows if ismf(r['Kernel_Name'])]
launches = []
for line in open(f'{O}/pl_trace{B}.log', errors='ignore'):
    m = re.search(r'EOSVOS_TRACE (\w+) conv=(\d+) M=(\d+) N=(\d+) K=(\d+) splits=(\d+) flops=(\d+)', line)
    if m: launches.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), float(m.group(7))))
assert len(launches) == len(mf), (len(launches), len(mf))
starts = [i for i, l in enumerate(launches) if l[0] == 'fwd' and l[1] == 1]
per = starts[1] - starts[0]
s = starts[-2]
print(f'{"#":>3} {"kind":6} {"conv":>4} {"M":>7} {"N":>6} {"K":>6} {"WGs":>5} {"us":>7} {"TF/s":>6} | {"wave_cyc/1e6":>10} {"wait_any%":>9} {"wait_inst%":>10} {"active%":>8} {"valu%":>6} {"lds%":>5} {"mfma_busy% of CU-cycles":>12}')
for i in range(s, s + per):
    k, ci, M, N, K, sp, fl = launches[i]
    r = mf[i]; c = cnt[r['Dispatch_Id']]
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    # SQ_* wave counters are in quad-cycles summed over waves; MFMA busy in cycles summed over SIMDs(?) -> relative to busy cycles
    busy = c.get('SQ_BUSY_CYCLES', 0) or 1
    print(f'{i - s:3d} {k:6} {ci:4d} {M:7d} {N:6d} {K:6d} {int(r["Grid_Size_X"]) // 256:5d} {us:7.1f} {fl / us / 1e6:6.1f} | {wc / 1e6:10.2f} '
          f'{100 * c.get("SQ_WAIT_ANY", 0) / wc:9.1f} {100 * c.get("SQ_WAIT_INST_ANY", 0) / wc:10.1f} {100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc:8.1f} '
          f'{100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc:6.1f} {100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc:5.1f} {c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy:12.3f}')
PY
rm -rf $O/pl$B $O/pl_trace$B.log
head -70 $O/pmc_layers_b$B.txt
