#!/bin/bash
# where in an iteration is the chip under-filled?  For every interval of the two-queue kernel trace: the kernels running and their
# workgroups; prints the time spent with fewer than 256 workgroups in flight, by kernel name.   tools/debug/step_idle.sh [batch]
export TMPDIR=/tmp
B=${1:-3}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/step_profile.py $B > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pads = [i for i, r in enumerate(rows) if "nchw_to_nhwc_pad" in r["Kernel_Name"]]
i0, i1 = pads[-2], pads[-1]
seg = rows[i0:i1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[i1]["Start_Timestamp"])
ev = []
for r in seg:
    wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1))
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("eosvos::", "")
    ev.append((int(r["Start_Timestamp"]), 1, nm, wg)); ev.append((min(int(r["End_Timestamp"]), t1), -1, nm, wg))
ev.sort()
active = collections.Counter(); last = t0
low = collections.Counter(); idle = 0; lowtot = 0
for t, d, nm, wg in ev:
    dt = t - last
    if dt > 0:
        tot = sum(w for (n, w), c in active.items() for _ in range(c))
        if not active or sum(active.values()) == 0: idle += dt
        elif tot < 256:
            lowtot += dt
            for (n, w), c in active.items():
                if c > 0: low[n] += dt / sum(active.values())
    last = t
    active[(nm, wg)] += d
    if active[(nm, wg)] == 0: del active[(nm, wg)]
print("iteration %.1f us; nothing running %.1f us; fewer than 256 workgroups in flight %.1f us" % ((t1 - t0) / 1e3, idle / 1e3, lowtot / 1e3))
for n, v in low.most_common(16): print("  %-46s %7.1f us" % (n[:46], v / 1e3))
PY
rm -rf gpurun_out/tl
