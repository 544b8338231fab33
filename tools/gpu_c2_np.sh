# developer utility: C2 with one chunk per G2P wave (default below 1.5 M particles) against two (WGS_DEBUG = 131072)
for dbg in 0 131072 0 131072; do
  WGS_DEBUG=$dbg timeout 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra --allow-debug-switches --config c2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2 dbg=$dbg', round(d['ms_per_step']*1e3,1), {a:round(b*1e3,1) for a,b in d['pass_ms_per_step'].items() if b>0.0045})"
done
