# Developer utility: A/B of library variants (tools/tmp_libs/*.so) x environment sets on one box.
# usage: ENVS="WGS_DEBUG=2048|" BENCHARGS="" bash tools/gpu_lib_env_ab.sh
cd $GRAFT_REPO_ROOT
cp wgsparkl_amd/csrc/libwgsparkl3d_hip.so /tmp/orig.so
IFS='|' read -ra SETS <<< "${ENVS:-|}"
for rep in 1 2; do
for f in tools/tmp_libs/*.so; do
  cp $f wgsparkl_amd/csrc/libwgsparkl3d_hip.so
  for s in "${SETS[@]}" ""; do
    echo "== $f [$s]"; env $s timeout 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline $BENCHARGS 2>&1 | grep -o '"value": [0-9.]*\|"g2p": [0-9.]*\|"p2g": [0-9.]*\|"grid sort": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
  done
done; done
cp /tmp/orig.so wgsparkl_amd/csrc/libwgsparkl3d_hip.so
