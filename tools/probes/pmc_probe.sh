cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02b/pmc; rm -rf $O; mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/probes/bin/bf16x6_probe
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2","p3"):
    fs=glob.glob("$O/%s/**/*counter_collection.csv"%p,recursive=True)
    if not fs: print(p,"no csv"); continue
    acc=collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k=(r["Kernel_Name"][:40], r["Grid_Size"], r["Counter_Name"])
        acc.setdefault(k,[]).append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(p,k, "n=%d last=%.4g"%(len(v),v[-1]))
PY
