# developer utility: the mixed G2P launch (default) against the paired one (WGS_DEBUG = 262144), event-timed
for cfg in c2 c3 c5; do for dbg in 0 262144; do
  WGS_DEBUG=$dbg timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --allow-debug-switches --config $cfg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$cfg dbg=$dbg', round(d['ms_per_step']*1e3,1), {a:round(b*1e3,1) for a,b in d['pass_ms_per_step'].items() if b>0.0045})"
done; done
