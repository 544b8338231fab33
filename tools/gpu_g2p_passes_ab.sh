# developer utility: fused G2P with 1 / 2 chunks per wave at several sizes (event-timed G2P pass, us)
for P in 1 2; do
  WGS_EXTRA_FLAGS=-DWGS_G2P_PASSES=$P bash wgsparkl_amd/csrc/build.sh force
  for cfg in "--n-side 100" "--n-side 160" "--config c5"; do
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra $cfg 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('passes $P', '$cfg', 'substep', round(d['ms_per_step']*1e3,1), 'g2p', round(d['pass_ms_per_step']['g2p']*1e3,1), 'frac', round(d['roofline']['frac'],3))"
  done
done
