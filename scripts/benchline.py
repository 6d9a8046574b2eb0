"""stdin: bench.py output; prints ms/step, roofline.frac, isolated frac / average launch (argv[1] = label)."""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
iso = d['roofline'].get('isolated') or {}
print(sys.argv[1] if len(sys.argv) > 1 else '', '%.2f ms/step' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'],
      'isolated %.3f' % iso.get('frac', float('nan')), 'avg launch %.1f us' % iso.get('avg_launch_us', float('nan')))
