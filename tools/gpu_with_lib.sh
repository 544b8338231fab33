# developer utility: run CMD with ${LIBDIR:-tools/tmp_libs}/$LIB.so standing in for the 3D library
cp wgsparkl_amd/csrc/libwgsparkl3d_hip.so /tmp/keep.so
cp ${LIBDIR:-tools/tmp_libs}/$LIB.so wgsparkl_amd/csrc/libwgsparkl3d_hip.so
timeout ${TMO:-300} $CMD
cp /tmp/keep.so wgsparkl_amd/csrc/libwgsparkl3d_hip.so
