for rl in 100 250 300 75,100,150,200,250,300; do python bench.py --steps 3 --warmup 2 --no-cpu --no-e2e --windows 2 --read-len $rl --batch 2000000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$rl', 'ms_per_step %.3f' % d['ms_per_step'], 'value %.1f M' % (d['value'] / 1e6), 'median %.1f' % (d['value_median_of_windows']/1e6))"; done
