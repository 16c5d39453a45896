#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats, then PMC passes in separate runs
# (gpurun refuses --pmc together with sys/runtime traces).  Output under gpurun_out/prof_<tag>/.
set -o pipefail
TAG=${1:-r01}
ARGS=${2:-"--no-cpu --no-e2e --steps 10 --warmup 2"}  # (--no-e2e: the streamed leg launches the same kernel over batches of 2 M reads, which would mix two launch sizes into one average)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py $ARGS > $OUT/kt_bench.json 2> $OUT/kt_bench.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch_bench.json 2> $OUT/fetch_bench.err || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py $ARGS > $OUT/write_bench.json 2> $OUT/write_bench.err || exit 3
find $OUT -name '*.csv' | head -30
ls -la $OUT/*
