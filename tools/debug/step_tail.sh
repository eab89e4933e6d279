#!/bin/bash
# timeline (two streams) of the kernels around the end of an iteration:  tools/debug/step_tail.sh [batch]
export TMPDIR=/tmp
B=${1:-3}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/step_profile.py $B > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pads = [i for i, r in enumerate(rows) if "nchw_to_nhwc_pad" in r["Kernel_Name"]]
i1 = pads[-1]; i0 = pads[-2]
t0 = int(rows[i0]["Start_Timestamp"])
print("iteration: %.1f us, %d kernels" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, i1 - i0))
qs = sorted(set(r["Queue_Id"] for r in rows[i0:i1]))
for r in rows[i1 - 22:i1 + 4]:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("eosvos::", "")[:44]
    print("q%d %-44s start %8.1f  dur %7.1f" % (qs.index(r["Queue_Id"]) if r["Queue_Id"] in qs else 9, nm, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf gpurun_out/tl
