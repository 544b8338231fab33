cp wgsparkl_amd/csrc/libwgsparkl3d_hip.so /tmp/keep.so
for f in tools/tmp_libs/abl_*.so; do
  cp $f wgsparkl_amd/csrc/libwgsparkl3d_hip.so; echo "== $f"
  timeout 300 python tools/gpu_sort_prof.py c3 2>&1 | tail -10
done
cp /tmp/keep.so wgsparkl_amd/csrc/libwgsparkl3d_hip.so
