#!/bin/bash
# SQ instruction-mix counters for the classify kernels (separate PMC run, kernel-trace only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $ROOT/bench.py --no-cpu --no-e2e --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/b -o b -- python3 $ROOT/bench.py --no-cpu --no-e2e --steps 3 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv, collections
for t in ("a","b"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for r in csv.DictReader(open("$OUT/%s/%s_counter_collection.csv"%(t,t))):
            k=r["Kernel_Name"].split("(")[0][:64]
            if "classify_kernel<160" in k or "k4_" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e: print("ERR",e)
    for k,v in agg.items():
        print(k)
        for c,vals in v.items(): print("   %-24s %.4g"%(c, sum(vals)/len(vals)))
PY
