#!/bin/bash
# A/B of built libraries on the headline workload: scripts/ab_libs.sh lib1 lib2 ... (paths relative to the repo root); extra bench args in $BENCH_ARGS
for lib in "$@"; do LMAT_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --windows 2 --no-cands $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$lib', 'classify_ms %.3f' % d['roofline']['kernel_avg_ms'], 'step/launch %.3f' % d['roofline']['step_ms_per_launch'], 'value %.1f M' % (d['value'] / 1e6), 'median %.1f' % (d['value_median_of_windows']/1e6))"; done
