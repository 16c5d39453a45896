#!/bin/bash
# decision kernels after each part (debug switch LMAT_STOP_AFTER=10..12): per-kernel average from a kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for s in ${STOPS:-10 14 13 11 12 0}; do
  OUT=$ROOT/gpurun_out/k4ab/s$s; mkdir -p $OUT
  LMAT_STOP_AFTER=$s rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --no-cpu --no-e2e --steps 4 --warmup 1 > /dev/null 2>&1
  python3 - <<PY
import csv
print("stop=$s", " ".join("%s=%.2fms" % (r["Name"].split("(")[0].replace("void lmat::", "").replace("lmat::", ""), float(r["AverageNs"]) / 1e6) for r in csv.DictReader(open("$OUT/p_kernel_stats.csv")) if "k4_" in r["Name"]))
PY
done
