#!/bin/bash
# quick per-kernel timing: rocprofv3 kernel trace + stats of a short bench run -> gpurun_out/kt_<tag>.csv
TAG=${1:-x}
ARGS=${2:-"--no-cpu --steps 6 --warmup 2"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err || exit 1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $f $ROOT/gpurun_out/kt_$TAG.csv
