#!/bin/bash
# any counter set on the headline bench, classify kernel only: scripts/pmc_set.sh <lib> <tag> "<counters>"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$2; mkdir -p $OUT
LMAT_LIB=$ROOT/$1 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --no-cpu --no-e2e --windows 0 --steps 2 --warmup 1 > /dev/null 2>$OUT/err.txt
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "classify_kernel<160, 64, 256" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$2", " ".join("%s=%.4g"%(k, sum(v)/len(v)) for k,v in sorted(agg.items())))
PY
