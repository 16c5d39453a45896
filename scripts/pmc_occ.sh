#!/bin/bash
# average resident waves of the classify kernel: scripts/pmc_occ.sh <lib> <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_occ_$2; mkdir -p $OUT
LMAT_LIB=$ROOT/$1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --no-cpu --no-e2e --windows 0 --steps 2 --warmup 1 > /dev/null 2>$OUT/err.txt
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "classify_kernel<160, 64, 256" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$2", " ".join("%s=%.4g"%(k, sum(v)/len(v)) for k,v in sorted(agg.items())))
dur=[]
for r in csv.DictReader(open("$OUT/p_kernel_trace.csv")):
    if "classify_kernel<160, 64, 256" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
print("$2 kernel ms avg %.3f min %.3f n %d" % (sum(dur)/len(dur), min(dur), len(dur)))
PY
