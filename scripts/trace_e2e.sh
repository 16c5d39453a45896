#!/bin/bash
# kernel + memory-copy timeline of the boundary-inclusive bench leg (no counters): gpurun_out/trace_e2e/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/trace_e2e
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/gpurun_out/trace_e2e -o t -- python3 $ROOT/bench.py --no-cpu --steps 6 --warmup 1 > /dev/null 2>&1
cd $ROOT/gpurun_out/trace_e2e && python3 - <<PY
import csv
k=list(csv.DictReader(open("t_kernel_trace.csv")))
m=list(csv.DictReader(open("t_memory_copy_trace.csv")))
big=[r for r in m if int(r["End_Timestamp"])-int(r["Start_Timestamp"])>1000000]
t0=min(int(r["Start_Timestamp"]) for r in big)
ev=[]
for r in k:
    s=(int(r["Start_Timestamp"])-t0)/1e6; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if s>20 and d>0.05: ev.append((s,d,"K "+r["Kernel_Name"][:44]))
for r in m:
    s=(int(r["Start_Timestamp"])-t0)/1e6; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if s>20 and d>0.05: ev.append((s,d,"C "+r["Direction"]))
for e in sorted(ev)[:70]: print("%8.3f %7.3f %s"%e)
PY
