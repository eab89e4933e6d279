#!/bin/bash
export TMPDIR=/tmp EOSVOS_TUNE_STREAM3X3_MINM=0 EOSVOS_TUNE_STREAM1X1_MINM=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s3 -- python3 tools/debug/stream_time.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
rows=[r for r in csv.DictReader(open(glob.glob("gpurun_out/s3/**/*kernel_trace.csv",recursive=True)[0])) if "stream_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for i in range(0,len(rows),8):
    grp=rows[i:i+8]
    if len(grp)<8: break
    nm=lambda r: r["Kernel_Name"].split("eosvos::")[1].split("(")[0]
    f=min((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in grp[0::2]); d=min((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in grp[1::2])
    print(f"{nm(grp[0]):36s} {nm(grp[1]):36s} grid {grp[0]['Grid_Size_X']:>7s}  fwd {f:6.1f} us  dgrad {d:6.1f} us")
PY
rm -rf gpurun_out/s3
