for w in 5 50 200; do for rep in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup $w rep $rep: %.3f M  ms/step %.3f  steady %.3f M' % (d['value']/1e6, d['ms_per_step'], d['steady_state']['value']/1e6))"
done; done
