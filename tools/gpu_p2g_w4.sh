# developer utility: P2G pair at 4 waves per SIMD (128 VGPRs) against the shipped 3 (168)
sed -i 's/k_p2g_pair<D, false, 3>/k_p2g_pair<D, false, 4>/' wgsparkl_amd/csrc/capi.hip
bash wgsparkl_amd/csrc/build.sh force
bash tools/gpu_kstats.sh 2>&1 | grep -E "p2g|g2p" | cut -c1-150
