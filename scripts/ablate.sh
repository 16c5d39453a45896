#!/bin/bash
# phase ablation of the classify kernel (debug switch LMAT_STOP_AFTER): kernel ms per 1M reads after each phase
# Stops 1, 3, 5, 6, 30..34 and 40..47 exist in ablation builds only (the production kernel carries no checks for them):
#   scripts/build_variant.sh ablate lmat_amd/csrc/kernels.hip -DLMAT_ABLATE=1 && LMAT_LIB=$PWD/lmat_amd/variants/ablate.so <this script>
for s in ${STOPS:-1 2 3 4 5 6 0}; do
  LMAT_STOP_AFTER=$s python bench.py --db-gb ${1:-64} --batch 1000000 --steps 3 --warmup 1 --no-cpu 2>&1 >/dev/null | grep "timed region" | sed "s/^/stop_after=$s /"
done
