# developer utility: A/B on ONE box, alternating: every library under tools/tmp_libs/ and the in-tree one ("tree"), REPS times,
# event-timed passes of CFGS (default "c2 c3 c5")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp wgsparkl_amd/csrc/libwgsparkl3d_hip.so /tmp/tree.so
for rep in $(seq 1 ${REPS:-2}); do
  for f in tools/tmp_libs/*.so /tmp/tree.so; do
    cp $f wgsparkl_amd/csrc/libwgsparkl3d_hip.so
    for cfg in ${CFGS:-c2 c3 c5}; do
      timeout 200 python bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-extra --config $cfg $ARGS 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $f .so) $cfg', round(d['ms_per_step']*1e3,1), {a:round(b*1e3,1) for a,b in d['pass_ms_per_step'].items() if b>0.0045})"
    done
  done
done
cp /tmp/tree.so wgsparkl_amd/csrc/libwgsparkl3d_hip.so
