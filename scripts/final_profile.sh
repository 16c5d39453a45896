#!/bin/bash
# End-of-round measurement pass on the GPU box: headline profile (kernel trace + FETCH/WRITE counters), per-phase SQ counters
# (ablation build), the default bench line, BASELINE config 5 at its own table size, the heavy-tail leg.  Output: gpurun_out/<tag>/.
TAG=${1:-r04f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
bash scripts/profile_gpu.sh $TAG > $OUT/profile_gpu.log 2>&1; echo "profile rc $?"
LMAT_LIB=$ROOT/lmat_amd/variants/ablate.so STOPS="1 2 3 4 5 6 30 31 32 33 0" bash scripts/pmc_ablate.sh > $OUT/pmc_phases.txt 2>&1; echo "phases rc $?"; tail -1 $OUT/pmc_phases.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
python bench.py --db-gb 186 --read-len 75,100,125,150,200,250,300 --steps 10 --warmup 2 --no-cpu > $OUT/cfg5.json 2> $OUT/cfg5.err; echo "cfg5 rc $?"
python bench.py --list-tail 20,10,5 --steps 3 --warmup 1 --no-cpu --no-e2e > $OUT/heavy_tail.json 2> $OUT/heavy_tail.err; echo "tail rc $?"
python bench.py --list-tail 20,10,5 --prune 64 --steps 3 --warmup 1 --no-cpu --no-e2e > $OUT/heavy_tail_g64.json 2> $OUT/heavy_tail_g64.err; echo "tail-g64 rc $?"
echo done
