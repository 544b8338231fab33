# developer utility: event-timed passes of c2 at several sizes (SIDES) for every library variant under tools/tmp_libs/
cp wgsparkl_amd/csrc/libwgsparkl3d_hip.so /tmp/keep.so
for f in tools/tmp_libs/*.so /tmp/keep.so; do
  cp $f wgsparkl_amd/csrc/libwgsparkl3d_hip.so
  for n in ${SIDES:-126 160}; do
  timeout 120 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --config c2 --n-side $n 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$f n_side=$n', round(d['ms_per_step']*1e3,1), {a:round(b*1e3,1) for a,b in d['pass_ms_per_step'].items() if b>0.0045})"
  done
done
cp /tmp/keep.so wgsparkl_amd/csrc/libwgsparkl3d_hip.so
