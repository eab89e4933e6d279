#!/bin/bash
# PMC passes on the bench's roofline kernel (one counter group per pass, gpurun-safe: --pmc with --kernel-trace only)
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 tools/hot_kernel.py > $O/$c.log 2>&1; tail -1 $O/$c.log
  python3 - <<PY
import csv,glob
f=glob.glob("$O/$c/**/*counter_collection.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "conv_igemm_kernel<128, false, 2>" in r["Kernel_Name"] and r["Counter_Name"]=="$c" and r["Grid_Size"]=="196608"]
vals=[float(r["Counter_Value"]) for r in rows]
print("$c (KB per launch, last 5 768-workgroup launches of conv_igemm_kernel<128,false,2> = Winograd GEMM of decoder.last_conv.0 forward):", vals[-5:])
PY
done
