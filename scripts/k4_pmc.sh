#!/bin/bash
# instruction counts of the decision kernels per wave, after scores (stop 14) and after the sort (stop 13)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for s in 14 13 0; do
  OUT=$ROOT/gpurun_out/k4pmc/s$s; mkdir -p $OUT
  LMAT_STOP_AFTER=$s rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --no-cpu --no-e2e --steps 2 --warmup 1 > /dev/null 2>&1
  python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void lmat::", "")
    if "k4_" in k and "compact" not in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w = max(m.get("SQ_WAVES", 1), 1)
    print("stop=$s %-28s waves %6.0f  per wave: VALU %7.0f SALU %7.0f LDS %6.0f FLAT %6.0f  cycles(x4) %8.0f wait %8.0f active %8.0f" % (k, w, m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_FLAT", 0) / w, m.get("SQ_WAVE_CYCLES", 0) / w, m.get("SQ_WAIT_ANY", 0) / w, m.get("SQ_ACTIVE_INST_ANY", 0) / w))
PY
done
