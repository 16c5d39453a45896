#!/bin/bash
# instruction counts of the classify kernel after each phase (debug switch LMAT_STOP_AFTER)
# Stops 1, 3, 5, 6, 30..34 and 40..47 exist in ablation builds only (the production kernel carries no checks for them):
#   scripts/build_variant.sh ablate lmat_amd/csrc/kernels.hip -DLMAT_ABLATE=1 && LMAT_LIB=$PWD/lmat_amd/variants/ablate.so <this script>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for s in ${STOPS:-1 2 3 4 5 6 0}; do
  OUT=$ROOT/gpurun_out/pmc_ab/s$s; mkdir -p $OUT
  LMAT_STOP_AFTER=$s rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --no-cpu --no-e2e --windows 0 --steps 2 --warmup 1 --batch 1000000 $BENCH_EXTRA > /dev/null 2>&1
  python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "classify_kernel<160, 64, 256" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("stop=$s", " ".join("%s=%.0f"%(k.replace("SQ_INSTS_","").replace("SQ_",""), sum(v)/len(v)/1e6) for k,v in sorted(agg.items())), "(per read, 1M reads)")
PY
done
