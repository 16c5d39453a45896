#!/bin/bash
# A/B two builds of the library inside ONE gpurun call (boxes differ by up to 2x, so never compare across calls)
ARGS=${ARGS:-"--no-cpu --steps 6 --warmup 2"}
for rep in 1 2; do
for lib in "$@"; do
  LMAT_LIB=$PWD/lmat_amd/$lib python bench.py $ARGS 2>&1 | grep -E "timed region|db built" | sed "s/^/$lib /"
done
done
