# developer utility: event-timed passes of the bench configurations (CFGS, default "c2 c3 c5")
for cfg in ${CFGS:-c2 c3 c5}; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --config $cfg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step']*1e3,1), {a:round(b*1e3,1) for a,b in d['pass_ms_per_step'].items() if b>0.0045})"
done
