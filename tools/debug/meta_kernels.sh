#!/bin/bash
# kernel-time breakdown of the meta path at N tasks per rank:  tools/debug/meta_kernels.sh [tasks_per_rank]
export TMPDIR=/tmp
T=${1:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mp -- python3 bench.py --metric meta --tasks-per-rank $T --steps 20 --warmup 3 --no-cpu-baseline --no-ab > gpurun_out/meta_tpr$T.json 2>/dev/null
python3 - "$T" <<'PY'
import csv, glob, json, sys
f = glob.glob("gpurun_out/mp/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 2))
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:40]:
    print("%-72s %7s %9.2f ms %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
d = json.loads(open("gpurun_out/meta_tpr%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("tasks/s", d["value"], "ms per meta-iteration", d["ms_per_step"])
PY
rm -rf gpurun_out/mp
