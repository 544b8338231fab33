cd $GRAFT_REPO_ROOT
cp wgsparkl_amd/csrc/libwgsparkl3d_hip.so /tmp/orig.so
for f in tools/tmp_libs/*.so; do
  cp $f wgsparkl_amd/csrc/libwgsparkl3d_hip.so
  echo "== $f"; timeout 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline ${VARARGS---no-floor} 2>&1 | grep -o '"value": [0-9.]*\|"g2p": [0-9.]*\|"p2g": [0-9.]*\|"grid sort": [0-9.]*' | tr '\n' ' '; echo
done
cp /tmp/orig.so wgsparkl_amd/csrc/libwgsparkl3d_hip.so
