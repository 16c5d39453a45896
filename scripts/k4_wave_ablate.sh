#!/bin/bash
# Where the decision step on the wave spends its time: bench runs that end k4_wave early (LMAT_STOP_AFTER 30..33, kernels.hip).
# Stops 1, 3, 5, 6, 30..34 and 40..47 exist in ablation builds only (the production kernel carries no checks for them):
#   scripts/build_variant.sh ablate lmat_amd/csrc/kernels.hip -DLMAT_ABLATE=1 && LMAT_LIB=$PWD/lmat_amd/variants/ablate.so <this script>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for s in 30 31 32 33 0; do
  LMAT_STOP_AFTER=$s python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --windows 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('stop', $s, 'classify_ms %.3f' % d['roofline']['kernel_avg_ms'], 'value %.1f M' % (d['value'] / 1e6))"
done
