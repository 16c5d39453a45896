#!/bin/bash
# same-box A/B of library builds: lmat_amd/variants/<name>.so are swapped in turn (boxes differ by a few per cent, so
# variants are only comparable inside one gpurun call); prints classify / decide ms per batch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cp $ROOT/lmat_amd/liblmat_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp $ROOT/lmat_amd/variants/$v.so $ROOT/lmat_amd/liblmat_hip.so
    python $ROOT/bench.py --no-cpu --no-e2e ${BENCH_ARGS:-} 2>&1 >/dev/null | grep "timed region" | sed "s/^/$v rep$rep /"
  done
done
cp /tmp/lib_keep.so $ROOT/lmat_amd/liblmat_hip.so
