#!/bin/bash
# per-kernel times of a short headline bench run: scripts/prof_quick.sh <lib> <tag> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIB=$1; TAG=$2; shift 2
LMAT_LIB=$R/$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu --no-e2e --windows 0 "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
f=$(find $R/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    if "synth" in r["Name"] or "cpt_" in r["Name"] or "fillBuffer" in r["Name"]: continue
    print("%-70s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
