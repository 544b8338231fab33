# developer utility: a launch-shape switch (DBG, WGS_DEBUG bits) against the default, event-timed, for CFGS
for rep in 1 2; do for cfg in ${CFGS:-c5 c2}; do for dbg in 0 ${DBG:-4096}; do
  WGS_DEBUG=$dbg timeout 120 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --allow-debug-switches --config $cfg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$cfg dbg=$dbg', round(d['ms_per_step']*1e3,1), {a:round(b*1e3,1) for a,b in d['pass_ms_per_step'].items() if b>0.0045})"
done; done; done
