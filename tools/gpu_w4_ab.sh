# developer utility: A/B of the G2P register budget (3 vs 4 waves per SIMD)
echo "== 3 waves (shipped), no floor"; ARGS="--no-floor" bash tools/gpu_kstats.sh 2>&1 | grep -E "g2p" | cut -c1-150
echo "== rebuild with 4 waves"; WGS_EXTRA_FLAGS=-DG2P_WAVES_PER_EU=4 bash wgsparkl_amd/csrc/build.sh force
echo "== 4 waves, floor"; bash tools/gpu_kstats.sh 2>&1 | grep -E "g2p" | cut -c1-150
echo "== 4 waves, no floor"; ARGS="--no-floor" bash tools/gpu_kstats.sh 2>&1 | grep -E "g2p" | cut -c1-150
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d['roofline']['frac'])
for k,v in d['extra'].items(): print(k, round(v['ms_per_step']*1e3,1), round(v['roofline_g2p']['frac'],3), round(v['pass_ms_per_step']['g2p']*1e3,1))"
