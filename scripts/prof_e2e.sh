#!/bin/bash
# per-kernel times of a bench run WITH the streamed leg (pack kernel, copies): scripts/prof_e2e.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu --windows 0 > $R/gpurun_out/prof_$1.json 2> $R/gpurun_out/prof_$1.err
f=$(find $R/gpurun_out/prof_$1 -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    if "synth" in r["Name"] or "cpt_" in r["Name"]: continue
    print("%-70s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
