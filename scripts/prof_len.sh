#!/bin/bash
# per-kernel times of a bench run at another read length: scripts/prof_len.sh <read-len> <lib> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LMAT_LIB=$R/$2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$3 -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu --no-e2e --windows 0 --read-len $1 --batch 2000000 > $R/gpurun_out/prof_$3.json 2> $R/gpurun_out/prof_$3.err
f=$(find $R/gpurun_out/prof_$3 -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
