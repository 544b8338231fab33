# developer utility: the default bench line (all extra legs, no CPU baseline) for every library under tools/tmp_libs/ and the in-tree one, on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp wgsparkl_amd/csrc/libwgsparkl3d_hip.so /tmp/tree.so
for f in ${LIBS:-tools/tmp_libs/*.so} /tmp/tree.so; do
  cp $f wgsparkl_amd/csrc/libwgsparkl3d_hip.so
  timeout 900 python bench.py --no-cpu-baseline --no-live-pmc 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$(basename $f .so)', 'c2', round(d['ms_per_step']*1e3,1), {k: (round(v['ms_per_step']*1e3,1), round(v['roofline_g2p']['frac'],3)) for k,v in d['extra'].items()})"
done
cp /tmp/tree.so wgsparkl_amd/csrc/libwgsparkl3d_hip.so
