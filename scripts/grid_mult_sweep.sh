for g in 16 32 48 64; do LMAT_GRID_MULT=$g python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --windows 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('gmult $g', 'classify_ms %.3f' % d['roofline']['kernel_avg_ms'], 'step/launch %.3f' % d['roofline']['step_ms_per_launch'], 'value %.1f M' % (d['value'] / 1e6))"; done
