# developer utility: k_regroup register budget A/B at C2 and C3 (event-timed sort pass, us)
for W in 4 3 1; do
  WGS_EXTRA_FLAGS=-DWGS_REGROUP_WPE=$W bash wgsparkl_amd/csrc/build.sh force
  for cfg in "--config c2" "--config c3"; do
    timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra $cfg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('wpe $W', '$cfg', 'substep', round(d['ms_per_step']*1e3,1), 'sort', round(d['pass_ms_per_step']['grid sort']*1e3,1))"
  done
done
