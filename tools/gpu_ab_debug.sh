# developer utility: event-timed passes of bench configurations (CFGS) with and without WGS_DEBUG bits (BITS) on the in-tree library
for cfg in ${CFGS:-c2 c3 c5}; do
  for bits in 0 ${BITS:-262144}; do  # (BITS may hold several values)
  WGS_DEBUG=$bits timeout 200 python bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-extra --allow-debug-switches --config $cfg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$bits $cfg', round(d['ms_per_step']*1e3,1), {a:round(b*1e3,1) for a,b in d['pass_ms_per_step'].items() if b>0.0045})"
  done
done
